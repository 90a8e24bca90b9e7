// Embedding head and loss kernels: L2 row normalisation, the batch-wide queue
// hit mask, masked InfoNCE rows, label-smoothed cross-entropy rows, projection
// column normalisation, global-align softplus terms, small reductions.
// Reference: head.py:126-175, losses.py:6-62,102-128,206-217, moco_head/loss.py.
// The [rows, K] similarity / logit matrices are produced by trid_gemm_f32; the
// kernels here turn them into per-row losses and, in place, into dL/dlogits, so
// that the backward pass is two more GEMMs and no second softmax.

#include "common.h"

namespace trid {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int i = 0; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}

// one wave per row
__global__ void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm,
                                   long long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int j = lane; j < C; j += 64) s = fmaf(xr[j], xr[j], s);
    s = wave_sum(s);
    const float inv = 1.f / fmaxf(sqrtf(s), eps);
    for (int j = lane; j < C; j += 64) y[row * C + j] = xr[j] * inv;
    if (lane == 0 && inv_norm != nullptr) inv_norm[row] = inv;
}

__global__ void l2norm_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                       const float* __restrict__ inv_norm, float* __restrict__ dx, long long rows, int C,
                                       int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* dr = dy + row * C;
    const float* yr = y + row * C;
    float d = 0.f;
    for (int j = lane; j < C; j += 64) d = fmaf(dr[j], yr[j], d);
    d = wave_sum(d);
    const float inv = inv_norm[row];
    for (int j = lane; j < C; j += 64) {
        const float v = (dr[j] - yr[j] * d) * inv;
        dx[row * C + j] = accumulate ? dx[row * C + j] + v : v;
    }
}

__global__ void queue_hit_mask_kernel(const int64_t* __restrict__ id_queue, const int64_t* __restrict__ ids,
                                      uint8_t* __restrict__ flag, int K, int B) {
    extern __shared__ int64_t sid[];
    for (int i = threadIdx.x; i < B; i += blockDim.x) sid[i] = ids[i];
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int64_t v = id_queue[k];
    bool hit = false;
    for (int i = 0; i < B; ++i) hit |= (sid[i] == v);
    flag[k] = hit ? 1 : 0;
}

// InfoNCE over one query row's masked queue logits, split over the row so that long queues fill the chip:
// a row of K logits is cut into segments of INFONCE_SEG columns, one workgroup each.
//   pass 1 (infonce_partial_kernel): per segment, max and sum of exp over the unmasked columns;
//   pass 2 (infonce_finish_kernel) : every workgroup folds the row's segment partials (fixed order, so the
//          result does not depend on scheduling) with the positive logit into the row's log-sum-exp,
//          then rewrites its segment in place as dL/dS; segment 0 also emits the row loss and dL/dpos.
// float4 accesses throughout (K and ldS multiples of 4 take the vector path).
constexpr int INFONCE_SEG = 4096;

__global__ __launch_bounds__(256) void infonce_partial_kernel(const float* __restrict__ S, const uint8_t* __restrict__ hit,
                                                              float2* __restrict__ part, int K, int ldS, float invT) {
    __shared__ float red[8];
    const int b = blockIdx.x, seg = blockIdx.y, nseg = gridDim.y;
    const float* r = S + (long long)b * ldS;
    const int k0 = seg * INFONCE_SEG, k1 = min(K, k0 + INFONCE_SEG);
    const bool vec = ((ldS & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    float m = -INFINITY;
    if (vec) {
        for (int k = k0 + 4 * threadIdx.x; k < k1; k += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(r + k);
            const uchar4 h = *reinterpret_cast<const uchar4*>(hit + k);  // k % 4 == 0; tail handled below
            if (k + 3 < k1) {
                if (!h.x) m = fmaxf(m, v.x * invT);
                if (!h.y) m = fmaxf(m, v.y * invT);
                if (!h.z) m = fmaxf(m, v.z * invT);
                if (!h.w) m = fmaxf(m, v.w * invT);
            } else {
                for (int t = k; t < k1; ++t) if (!hit[t]) m = fmaxf(m, r[t] * invT);
            }
        }
    } else {
        for (int k = k0 + threadIdx.x; k < k1; k += 256) if (!hit[k]) m = fmaxf(m, r[k] * invT);
    }
    m = block_max(m, red);
    float l = 0.f;
    if (m > -INFINITY) {
        if (vec) {
            for (int k = k0 + 4 * threadIdx.x; k < k1; k += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(r + k);
                const uchar4 h = *reinterpret_cast<const uchar4*>(hit + k);
                if (k + 3 < k1) {
                    if (!h.x) l += expf(v.x * invT - m);
                    if (!h.y) l += expf(v.y * invT - m);
                    if (!h.z) l += expf(v.z * invT - m);
                    if (!h.w) l += expf(v.w * invT - m);
                } else {
                    for (int t = k; t < k1; ++t) if (!hit[t]) l += expf(r[t] * invT - m);
                }
            }
        } else {
            for (int k = k0 + threadIdx.x; k < k1; k += 256) if (!hit[k]) l += expf(r[k] * invT - m);
        }
    }
    l = block_sum(l, red);
    if (threadIdx.x == 0) part[(long long)b * nseg + seg] = make_float2(m, l);
}

// `pos` may be null: then the positive logit <q_b, key_b> is computed here from q / key [B, C] (saves a
// launch), and segment 0 also writes dq0[b, :] = dL/dpos * key_b, the positive-pair term of dL/dq that the
// gradient GEMM then accumulates onto.
__global__ __launch_bounds__(256) void infonce_finish_kernel(float* __restrict__ S, const float* __restrict__ pos,
                                                             const uint8_t* __restrict__ hit,
                                                             const float2* __restrict__ part,
                                                             float* __restrict__ loss_rows, float* __restrict__ dpos,
                                                             int K, int ldS, float invT, float gs,
                                                             const float* __restrict__ q, const float* __restrict__ key,
                                                             float* __restrict__ dq0, int C) {
    __shared__ float red[8];
    const int b = blockIdx.x, seg = blockIdx.y, nseg = gridDim.y;
    float* r = S + (long long)b * ldS;
    float praw;
    if (pos != nullptr) {
        praw = pos[b];
    } else {
        float d = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) d = fmaf(q[(long long)b * C + c], key[(long long)b * C + c], d);
        praw = block_sum(d, red);
    }
    const float p = praw * invT;
    float m = p;
    for (int s2 = 0; s2 < nseg; ++s2) m = fmaxf(m, part[(long long)b * nseg + s2].x);
    float l = expf(p - m);
    for (int s2 = 0; s2 < nseg; ++s2) {
        const float2 q = part[(long long)b * nseg + s2];
        if (q.x > -INFINITY) l += q.y * expf(q.x - m);
    }
    const float lse = m + logf(l);
    const int k0 = seg * INFONCE_SEG, k1 = min(K, k0 + INFONCE_SEG);
    const bool vec = ((ldS & 3) == 0) && ((reinterpret_cast<uintptr_t>(S) & 15) == 0);
    if (vec) {
        for (int k = k0 + 4 * threadIdx.x; k < k1; k += 1024) {
            if (k + 3 < k1) {
                float4 v = *reinterpret_cast<const float4*>(r + k);
                const uchar4 h = *reinterpret_cast<const uchar4*>(hit + k);
                v.x = h.x ? 0.f : expf(v.x * invT - lse) * gs;
                v.y = h.y ? 0.f : expf(v.y * invT - lse) * gs;
                v.z = h.z ? 0.f : expf(v.z * invT - lse) * gs;
                v.w = h.w ? 0.f : expf(v.w * invT - lse) * gs;
                *reinterpret_cast<float4*>(r + k) = v;
            } else {
                for (int t = k; t < k1; ++t) r[t] = hit[t] ? 0.f : expf(r[t] * invT - lse) * gs;
            }
        }
    } else {
        for (int k = k0 + threadIdx.x; k < k1; k += 256) r[k] = hit[k] ? 0.f : expf(r[k] * invT - lse) * gs;
    }
    if (seg == 0) {
        const float dp = (expf(p - lse) - 1.f) * gs;
        if (threadIdx.x == 0) {
            loss_rows[b] = lse - p;
            if (dpos != nullptr) dpos[b] = dp;
        }
        if (dq0 != nullptr)
            for (int c = threadIdx.x; c < C; c += 256) dq0[(long long)b * C + c] = dp * key[(long long)b * C + c];
    }
}

__global__ void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                              long long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float d = 0.f;
    for (int j = lane; j < C; j += 64) d = fmaf(x[row * C + j], y[row * C + j], d);
    d = wave_sum(d);
    if (lane == 0) out[row] = d;
}

__global__ void rowscale_add_kernel(const float* __restrict__ s, const float* __restrict__ y, float* __restrict__ dx,
                                    long long rows, int C, int accumulate) {
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const float v = s[i / C] * y[i];
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// one workgroup per row; in place logits -> dlogits
__global__ __launch_bounds__(256) void smooth_ce_rows_kernel(float* __restrict__ logits,
                                                             const int64_t* __restrict__ labels,
                                                             float* __restrict__ loss_rows, int n, int ld, float eps,
                                                             float gs) {
    __shared__ float red[8];
    const long long row = blockIdx.x;
    float* r = logits + row * ld;
    const int y = (int)labels[row];
    float m = -INFINITY, sum = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float v = r[j];
        m = fmaxf(m, v);
        sum += v;
    }
    m = block_max(m, red);
    sum = block_sum(sum, red);
    float l = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) l += expf(r[j] - m);
    l = block_sum(l, red);
    const float lse = m + logf(l);
    const float sy = r[y];
    __syncthreads();
    const float un = eps / (float)n;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float g = 0.f;
        if (j < n) g = (expf(r[j] - lse) - un - (j == y ? (1.f - eps) : 0.f)) * gs;
        r[j] = g;
    }
    if (threadIdx.x == 0) loss_rows[row] = -(1.f - eps) * (sy - lse) - un * (sum - (float)n * lse);
}

// projection [C,N] -> pn [C,ldn] (optional) and pnt [ldn,C]: one workgroup per 64 columns.
// Reads are coalesced along N, the transposed write goes through an LDS tile.
__global__ __launch_bounds__(256) void colnorm_kernel(const float* __restrict__ proj, float* __restrict__ pn,
                                                      float* __restrict__ pnt, float* __restrict__ inv_norm, int C, int N,
                                                      int ldn) {
    __shared__ float tile[64][65];
    __shared__ float part[4][64];
    __shared__ float invs[64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64;
    const int j = j0 + cl;
    float s = 0.f;
    if (j < N)
        for (int c = rl; c < C; c += 4) {
            const float v = proj[(long long)c * N + j];
            s = fmaf(v, v, s);
        }
    part[rl][cl] = s;
    __syncthreads();
    if (rl == 0) {
        const float t = part[0][cl] + part[1][cl] + part[2][cl] + part[3][cl];
        const float inv = j < N ? 1.f / fmaxf(sqrtf(t), 1e-12f) : 0.f;
        invs[cl] = inv;
        if (j < N) inv_norm[j] = inv;
    }
    __syncthreads();
    const float inv = invs[cl];
    for (int c0 = 0; c0 < C; c0 += 64) {
        for (int c = c0 + rl; c < c0 + 64 && c < C; c += 4) {
            const float v = j < N ? proj[(long long)c * N + j] * inv : 0.f;
            if (pn != nullptr && j < ldn) pn[(long long)c * ldn + j] = v;
            tile[c - c0][cl] = v;
        }
        __syncthreads();
        // write pnt rows j0..j0+63, columns c0..c0+63: lanes along c
        for (int jj = rl; jj < 64; jj += 4) {
            const int c = c0 + cl;
            if (j0 + jj < ldn && c < C) pnt[(long long)(j0 + jj) * C + c] = tile[cl][jj];
        }
        __syncthreads();
    }
}

// one wave per column j: dproj[c,j] = (dpnt[j,c] - pnt[j,c]*<dpnt_j,pnt_j>)*inv_j
__global__ void colnorm_bwd_kernel(const float* __restrict__ dpnt, const float* __restrict__ pnt,
                                   const float* __restrict__ inv_norm, float* __restrict__ dproj, int C, int N) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= N) return;
    const float* dr = dpnt + (long long)j * C;
    const float* pr = pnt + (long long)j * C;
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d = fmaf(dr[c], pr[c], d);
    d = wave_sum(d);
    const float inv = inv_norm[j];
    for (int c = lane; c < C; c += 64) dproj[(long long)c * N + j] = (dr[c] - pr[c] * d) * inv;
}

// one workgroup per row of the cosine matrix
__global__ __launch_bounds__(256) void global_align_rows_kernel(float* __restrict__ S, const int64_t* __restrict__ ids,
                                                                float* __restrict__ loss_rows, int B, int ldS,
                                                                float alpha, float beta, float sp, float sn, float gs) {
    __shared__ float red[8];
    const int i = blockIdx.x;
    float* r = S + (long long)i * ldS;
    const int64_t idi = ids[i];
    float acc = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) {
        const float s = r[j];
        float l, g;
        if (ids[j] == idi) {
            const float e = expf(-sp * (s - alpha));
            l = logf(1.f + e);
            g = -sp * e / (1.f + e);
        } else {
            const float e = expf(sn * (s - beta));
            l = logf(1.f + e);
            g = sn * e / (1.f + e);
        }
        acc += l;
        r[j] = g * gs;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) loss_rows[i] = acc * 2.f / (float)B;
}

__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, float* __restrict__ out, long long n,
                                                  float scale, int accumulate) {
    __shared__ float red[8];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) s += x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s * scale : s * scale;
}

// dx = act > 0 ? dy : 0   (ReLU backward of the MoCo head's fc branches, head.py:33-42)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act, float* __restrict__ dx, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] = act[i] > 0.f ? dy[i] : 0.f;
}

// out = g[0]*a + g[1]*b + g[2]*c with device-resident scalars (b, c optional)
__global__ void axpby3_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b,
                              const float* __restrict__ c, const float* __restrict__ g, long long n) {
    const float g0 = g[0], g1 = b ? g[1] : 0.f, g2 = c ? g[2] : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = g0 * a[i];
        if (b) v = fmaf(g1, b[i], v);
        if (c) v = fmaf(g2, c[i], v);
        out[i] = v;
    }
}

}  // namespace trid

using namespace trid;

extern "C" int trid_relu_bwd_f32(const float* dy, const float* act, float* dx, long long n, void* stream) {
    TRID_REQUIRE(dy && act && dx && n > 0, "trid_relu_bwd_f32: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n, 256 * 4, 2048)), dim3(256), 0, (hipStream_t)stream, dy, act, dx, n);
    return check_launch("trid_relu_bwd_f32");
}

extern "C" int trid_axpby3_f32(float* out, const float* a, const float* b, const float* c, const float* g3,
                               long long n, void* stream) {
    TRID_REQUIRE(out && a && g3 && n > 0, "trid_axpby3_f32: bad arguments");
    hipLaunchKernelGGL(axpby3_kernel, dim3(grid_for(n, 256 * 4, 2048)), dim3(256), 0, (hipStream_t)stream, out, a, b, c, g3,
                       n);
    return check_launch("trid_axpby3_f32");
}

extern "C" int trid_l2norm_rows_f32(const float* x, float* y, float* inv_norm, long long rows, int C, float eps,
                                    void* stream) {
    TRID_REQUIRE(x && y && rows > 0 && C > 0, "trid_l2norm_rows_f32: bad arguments");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y,
                       inv_norm, rows, C, eps);
    return check_launch("trid_l2norm_rows_f32");
}

extern "C" int trid_l2norm_rows_bwd_f32(const float* dy, const float* y, const float* inv_norm, float* dx,
                                        long long rows, int C, int accumulate, void* stream) {
    TRID_REQUIRE(dy && y && inv_norm && dx && rows > 0 && C > 0, "trid_l2norm_rows_bwd_f32: bad arguments");
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, y,
                       inv_norm, dx, rows, C, accumulate);
    return check_launch("trid_l2norm_rows_bwd_f32");
}

extern "C" int trid_queue_hit_mask(const int64_t* id_queue, const int64_t* ids, uint8_t* flag, int K, int B,
                                   void* stream) {
    TRID_REQUIRE(id_queue && ids && flag && K > 0 && B > 0 && B <= 8192, "trid_queue_hit_mask: bad arguments");
    hipLaunchKernelGGL(queue_hit_mask_kernel, dim3((K + 255) / 256), dim3(256), (size_t)B * sizeof(int64_t),
                       (hipStream_t)stream, id_queue, ids, flag, K, B);
    return check_launch("trid_queue_hit_mask");
}

extern "C" long long trid_infonce_ws_floats(int B, int K) { return 2LL * B * ((K + INFONCE_SEG - 1) / INFONCE_SEG); }

extern "C" int trid_infonce_rows_f32(float* S, const float* pos, const uint8_t* hit, float* loss_rows, float* dpos,
                                     int B, int K, int ldS, float invT, float gscale, float* ws, void* stream) {
    TRID_REQUIRE(S && pos && hit && loss_rows && dpos && ws && B > 0 && K > 0 && ldS >= K, "trid_infonce_rows_f32: bad arguments");
    TRID_REQUIRE((reinterpret_cast<uintptr_t>(hit) & 3) == 0 && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, "trid_infonce_rows_f32: hit must be 4-byte and ws 8-byte aligned");
    const int nseg = (K + INFONCE_SEG - 1) / INFONCE_SEG;
    hipLaunchKernelGGL(infonce_partial_kernel, dim3(B, nseg), dim3(256), 0, (hipStream_t)stream, S, hit, (float2*)ws, K, ldS, invT);
    hipLaunchKernelGGL(infonce_finish_kernel, dim3(B, nseg), dim3(256), 0, (hipStream_t)stream, S, pos, hit, (const float2*)ws,
                       loss_rows, dpos, K, ldS, invT, gscale * invT / (float)B, (const float*)nullptr, (const float*)nullptr,
                       (float*)nullptr, 0);
    return check_launch("trid_infonce_rows_f32");
}

extern "C" int trid_infonce_queue_rows_f32(float* S, const float* q, const float* key, const uint8_t* hit, float* loss_rows,
                                           float* dq0, int B, int K, int ldS, int C, float invT, float gscale, float* ws,
                                           void* stream) {
    TRID_REQUIRE(S && q && key && hit && loss_rows && dq0 && ws && B > 0 && K > 0 && C > 0 && ldS >= K,
                 "trid_infonce_queue_rows_f32: bad arguments");
    TRID_REQUIRE((reinterpret_cast<uintptr_t>(hit) & 3) == 0 && (reinterpret_cast<uintptr_t>(ws) & 7) == 0,
                 "trid_infonce_queue_rows_f32: hit must be 4-byte and ws 8-byte aligned");
    const int nseg = (K + INFONCE_SEG - 1) / INFONCE_SEG;
    hipLaunchKernelGGL(infonce_partial_kernel, dim3(B, nseg), dim3(256), 0, (hipStream_t)stream, S, hit, (float2*)ws, K, ldS, invT);
    hipLaunchKernelGGL(infonce_finish_kernel, dim3(B, nseg), dim3(256), 0, (hipStream_t)stream, S, (const float*)nullptr, hit,
                       (const float2*)ws, loss_rows, (float*)nullptr, K, ldS, invT, gscale * invT / (float)B, q, key, dq0, C);
    return check_launch("trid_infonce_queue_rows_f32");
}

extern "C" int trid_rowdot_f32(const float* x, const float* y, float* out, long long rows, int C, void* stream) {
    TRID_REQUIRE(x && y && out && rows > 0 && C > 0, "trid_rowdot_f32: bad arguments");
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, out,
                       rows, C);
    return check_launch("trid_rowdot_f32");
}

extern "C" int trid_rowscale_add_f32(const float* s, const float* y, float* dx, long long rows, int C, int accumulate,
                                     void* stream) {
    TRID_REQUIRE(s && y && dx && rows > 0 && C > 0, "trid_rowscale_add_f32: bad arguments");
    hipLaunchKernelGGL(rowscale_add_kernel, dim3(grid_for(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, s, y, dx,
                       rows, C, accumulate);
    return check_launch("trid_rowscale_add_f32");
}

extern "C" int trid_smooth_ce_rows_f32(float* logits, const int64_t* labels, float* loss_rows, long long rows, int n,
                                       int ld, float epsilon, float gscale, void* stream) {
    TRID_REQUIRE(logits && labels && loss_rows && rows > 0 && n > 0 && ld >= n, "trid_smooth_ce_rows_f32: bad arguments");
    hipLaunchKernelGGL(smooth_ce_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, labels,
                       loss_rows, n, ld, epsilon, gscale);
    return check_launch("trid_smooth_ce_rows_f32");
}

extern "C" int trid_colnorm_f32(const float* proj, float* pn, float* pnt, float* inv_norm, int C, int N, int ldn,
                                void* stream) {
    TRID_REQUIRE(proj && pnt && inv_norm && C > 0 && N > 0 && ldn >= N, "trid_colnorm_f32: bad arguments");
    hipLaunchKernelGGL(colnorm_kernel, dim3((ldn + 63) / 64), dim3(256), 0, (hipStream_t)stream, proj, pn, pnt, inv_norm, C,
                       N, ldn);
    return check_launch("trid_colnorm_f32");
}

extern "C" int trid_colnorm_bwd_f32(const float* dpnt, const float* pnt, const float* inv_norm, float* dproj, int C,
                                    int N, int ldn, void* stream) {
    TRID_REQUIRE(dpnt && pnt && inv_norm && dproj && C > 0 && N > 0 && ldn >= N, "trid_colnorm_bwd_f32: bad arguments");
    hipLaunchKernelGGL(colnorm_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, dpnt, pnt, inv_norm,
                       dproj, C, N);
    return check_launch("trid_colnorm_bwd_f32");
}

extern "C" int trid_global_align_rows_f32(float* S, const int64_t* ids, float* loss_rows, int B, int ldS, float alpha,
                                          float beta, float scale_pos, float scale_neg, float gscale, void* stream) {
    TRID_REQUIRE(S && ids && loss_rows && B > 0 && ldS >= B, "trid_global_align_rows_f32: bad arguments");
    hipLaunchKernelGGL(global_align_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, S, ids, loss_rows, B, ldS,
                       alpha, beta, scale_pos, scale_neg, gscale * 2.f / (float)B);
    return check_launch("trid_global_align_rows_f32");
}

extern "C" int trid_sum_f32(const float* x, float* out, long long n, float scale, int accumulate, void* stream) {
    TRID_REQUIRE(x && out && n > 0, "trid_sum_f32: bad arguments");
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, n, scale, accumulate);
    return check_launch("trid_sum_f32");
}
