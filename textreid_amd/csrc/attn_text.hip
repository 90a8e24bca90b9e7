// Attention-pool token kernels (m_resnet.py:103-135), row softmax, column sums,
// and the text-encoder kernels (gru.py:48-82): embedding-table gather, masked
// BiGRU cell forward/backward with the max-over-time fused in.
// The dense parts (projections, recurrent h @ W_hh^T) run on trid_gemm_f32.

#include "split_common.h"

namespace trid {

// tok[b,0,:] = mean_t x[b,t,:] + pos[0] ; tok[b,1+t,:] = x[b,t,:] + pos[1+t]
__global__ void attnpool_tokens_kernel(const float4* __restrict__ x, const float4* __restrict__ pos,
                                       float4* __restrict__ tok, int B, int T, int CQ, int ldt) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * CQ) return;
    const int cq = (int)(gid % CQ);
    const int b = (int)(gid / CQ);
    const float4* xb = x + (long long)b * T * CQ + cq;
    float4* tb = tok + (long long)b * ldt * CQ + cq;
    for (int t = T + 1; t < ldt; ++t) tb[(long long)t * CQ] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
        const float4 v = xb[(long long)t * CQ];
        const float4 p = pos[(long long)(t + 1) * CQ + cq];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        tb[(long long)(t + 1) * CQ] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
    }
    const float inv = 1.f / (float)T;
    const float4 p0 = pos[cq];
    tb[0] = make_float4(s.x * inv + p0.x, s.y * inv + p0.y, s.z * inv + p0.z, s.w * inv + p0.w);
}

// The same with the token loop spread over the workgroup (the kernel above walks a thread through all T tokens: 65 536
// threads, each a chain of T dependent iterations - 150-190 us for 200 MB) and the input read in the format the last
// residual block wrote it (IFMT 1: P16, 2: plain bf16; 0: fp32) - no unpack pass in front.  Block = 64 channel quads x 4
// token lanes; grid (C / 256, B).
template <int IFMT>
__global__ __launch_bounds__(256) void attnpool_tokens_wide_kernel(const float4* __restrict__ x, const float* __restrict__ x_amax,
                                                                   const float4* __restrict__ pos, float4* __restrict__ tok,
                                                                   int T, int CQ, int ldt) {
    __shared__ float4 red[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int cq = blockIdx.x * 64 + cl, b = blockIdx.y;
    const bool live = cq < CQ;
    const float inv_scale = IFMT == 1 ? 1.f / f16_scale_of(*x_amax) : 1.f;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        float4* tb = tok + (long long)b * ldt * CQ + cq;
        for (int t = tl; t < T; t += 4) {
            const long long i = ((long long)b * T + t) * CQ + cq;
            const float4 v = IFMT == 1 ? p16_load4(reinterpret_cast<const uint2*>(x), i, CQ, inv_scale)
                           : IFMT == 2 ? bf16_load4(reinterpret_cast<const uint2*>(x), i) : x[i];
            const float4 p = pos[(long long)(t + 1) * CQ + cq];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            tb[(long long)(t + 1) * CQ] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
        }
        for (int t = T + 1 + tl; t < ldt; t += 4) tb[(long long)t * CQ] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && live) {
        const float4 a = red[0][cl], c = red[1][cl], d = red[2][cl], e = red[3][cl];
        const float inv = 1.f / (float)T;
        const float4 p0 = pos[cq];
        tok[(long long)b * ldt * CQ + cq] = make_float4(((a.x + c.x) + (d.x + e.x)) * inv + p0.x, ((a.y + c.y) + (d.y + e.y)) * inv + p0.y,
                                                         ((a.z + c.z) + (d.z + e.z)) * inv + p0.z, ((a.w + c.w) + (d.w + e.w)) * inv + p0.w);
    }
}

__global__ void attnpool_tokens_bwd_dx_kernel(const float4* __restrict__ dtok, float4* __restrict__ dx, int B, int T,
                                              int CQ, int ldt) {
    const long long total = (long long)B * T * CQ;
    const float inv = 1.f / (float)T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % CQ);
        const long long bt = i / CQ;
        const int t = (int)(bt % T);
        const long long b = bt / T;
        const float4 a = dtok[(b * ldt + t + 1) * CQ + cq];
        const float4 m = dtok[(b * ldt) * CQ + cq];
        dx[i] = make_float4(a.x + m.x * inv, a.y + m.y * inv, a.z + m.z * inv, a.w + m.w * inv);
    }
}

__global__ void attnpool_tokens_bwd_dpos_kernel(const float4* __restrict__ dtok, float4* __restrict__ dpos, int B,
                                                int T1, int CQ, int ldt) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)T1 * CQ) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
        const float4 v = dtok[(long long)b * ldt * CQ + gid];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    dpos[gid] = s;
}

// one wave per row
__global__ void softmax_rows_kernel(float* __restrict__ s, long long rows, int n, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = s + row * ld;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, r[j]);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < n; j += 64) l += expf(r[j] - m);
    l = wave_sum(l);
    const float inv = 1.f / l;
    for (int j = lane; j < ld; j += 64) r[j] = j < n ? expf(r[j] - m) * inv : 0.f;
}

__global__ void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                        float* __restrict__ ds, long long rows, int n, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + row * ld;
    const float* dr = dp + row * ld;
    float* o = ds + row * ld;
    float d = 0.f;
    for (int j = lane; j < n; j += 64) d = fmaf(pr[j], dr[j], d);
    d = wave_sum(d);
    for (int j = lane; j < ld; j += 64) o[j] = j < n ? pr[j] * (dr[j] - d) : 0.f;
}

// out[n] (+)= sum_m x[m*ld+n]; block = 64 columns x 4 row lanes
__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long M, int N, long long ld,
                              int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (n < N)
        for (long long m = rl; m < M; m += 4) s += x[m * ld + n];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && n < N) {
        s = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
        out[n] = accumulate ? out[n] + s : s;
    }
}

// ------------------------------------------------------------------ text encoder
__global__ void embedding_gather_kernel(const float4* __restrict__ table, const int64_t* __restrict__ tokens,
                                        float4* __restrict__ x, int B, int L, int ldtok, int EQ, long long vocab) {
    const long long total = (long long)B * L * EQ;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % EQ);
        const long long bt = i / EQ;
        const int t = (int)(bt % L);
        const long long b = bt / L;
        long long id = tokens[b * ldtok + t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        x[i] = table[id * EQ + e];
    }
}

// Gradient of a TRAINABLE token-embedding table (gru.py:23-24: nn.Embedding(vocab, embed, padding_idx=0), the `use_onehot ==
// "yes"` form): dT[v] = sum of dX[p] over the positions p whose token is v, row `pad` stays zero.  Deterministic without a
// sort and without floating-point atomics: one workgroup per position; the workgroup of the FIRST position that holds a
// token owns that token's row and adds the rows of the later positions in position order.  dT is zero-filled by the caller.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dX, const int64_t* __restrict__ tokens, int ldtok,
                                                            int L, int N, int E, float* __restrict__ dT, long long vocab, long long pad) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x;
    const long long tok = tokens[(long long)(p / L) * ldtok + p % L];
    if (tok == pad || tok < 0 || tok >= vocab) return;
    __shared__ int earlier;
    __shared__ unsigned long long bal[4];
    if (tid == 0) earlier = 0;
    __syncthreads();
    for (int q = tid; q < p; q += 256)
        if (tokens[(long long)(q / L) * ldtok + q % L] == tok) earlier = 1;  // (any writer writes 1)
    __syncthreads();
    if (earlier) return;
    for (int c0 = 0; c0 < E; c0 += 256) {  // (E <= 256 * a few: the table rows are short)
        const int c = c0 + tid;
        float acc = c < E ? dX[(long long)p * E + c] : 0.f;
        for (int q0 = p + 1; q0 < N; q0 += 256) {
            const int q = q0 + tid;
            const bool hit = q < N && tokens[(long long)(q / L) * ldtok + q % L] == tok;
            const unsigned long long b = __ballot(hit);
            if (lane == 0) bal[wave] = b;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = bal[w];
                while (m) {
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (c < E) acc += dX[(long long)(q0 + 64 * w + j) * E + c];
                }
            }
            __syncthreads();
        }
        if (c < E) dT[tok * E + c] = acc;
    }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__global__ void gru_max_init_kernel(float* __restrict__ maxv, int32_t* __restrict__ argt,
                                    const int64_t* __restrict__ lengths, int Lmax, const int64_t* __restrict__ lmax_dev,
                                    int B, int Hd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * Hd) return;
    const int b = i / (2 * Hd);
    // pad_packed_sequence pads to the BATCH maximum (gru.py:78-79): that maximum is the host's Lmax, or - when the
    // host only knows an upper bound (a recorded step replayed on other captions) - the device scalar
    const long long batch_max = lmax_dev != nullptr ? lmax_dev[0] : (long long)Lmax;
    maxv[i] = lengths[b] < batch_max ? 0.f : -INFINITY;
    argt[i] = -1;
}

__global__ void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h,
                                    const int64_t* __restrict__ lengths, float* __restrict__ gates,
                                    float* __restrict__ hprev, float* __restrict__ maxv, int32_t* __restrict__ argt,
                                    int s, int Lmax, int L, int B, int Hd, long long gates_ds, long long hprev_ds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B * Hd) return;
    const int j = i % Hd;
    const int b = (i / Hd) % B;
    const int d = i / (Hd * B);
    const int t = d == 0 ? s : Lmax - 1 - s;
    const bool active = (long long)t < lengths[b];
    const float hp = h[i];
    if (hprev != nullptr) hprev[(long long)d * hprev_ds + (long long)b * Hd + j] = hp;
    if (!active) return;
    const float* gir = gi + ((long long)b * L + t) * (6 * Hd) + (long long)d * 3 * Hd;
    const float* ghr = gh + ((long long)d * B + b) * (3 * Hd);
    const float r = sigmoidf_(gir[j] + ghr[j]);
    const float z = sigmoidf_(gir[Hd + j] + ghr[Hd + j]);
    const float hn_lin = ghr[2 * Hd + j];
    const float n = tanhf(fmaf(r, hn_lin, gir[2 * Hd + j]));
    const float hnew = (1.f - z) * n + z * hp;
    h[i] = hnew;
    if (gates != nullptr) {
        float* gs = gates + (long long)d * gates_ds + (long long)b * (4 * Hd);
        gs[j] = r; gs[Hd + j] = z; gs[2 * Hd + j] = n; gs[3 * Hd + j] = hn_lin;
    }
    const int mc = b * 2 * Hd + d * Hd + j;
    // first-index-wins on ties, like torch.max(dim): forward walks t upward, reverse downward
    const float cur = maxv[mc];
    if (d == 0 ? (hnew > cur) : (hnew >= cur)) {
        maxv[mc] = hnew;
        argt[mc] = t;
    }
}

__global__ void gru_cell_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ argt,
                                    const float* __restrict__ gates, const float* __restrict__ hprev,
                                    const int64_t* __restrict__ lengths, float* __restrict__ dh,
                                    float* __restrict__ dGi, float* __restrict__ dgh, int s, int Lmax, int L, int B,
                                    int Hd, long long gates_ds, long long hprev_ds, long long dgh_ds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * B * Hd) return;
    const int j = i % Hd;
    const int b = (i / Hd) % B;
    const int d = i / (Hd * B);
    const int t = d == 0 ? s : Lmax - 1 - s;
    const bool active = (long long)t < lengths[b];
    float* dgir = dGi + ((long long)b * L + t) * (6 * Hd) + (long long)d * 3 * Hd;
    float* dghr = dgh + (long long)d * dgh_ds + (long long)b * (3 * Hd);
    if (!active) {
        dgir[j] = 0.f; dgir[Hd + j] = 0.f; dgir[2 * Hd + j] = 0.f;
        dghr[j] = 0.f; dghr[Hd + j] = 0.f; dghr[2 * Hd + j] = 0.f;
        return;
    }
    float dhv = dh[i];
    const int mc = b * 2 * Hd + d * Hd + j;
    if (argt[mc] == t) dhv += dout[mc];
    const float* gs = gates + (long long)d * gates_ds + (long long)b * (4 * Hd);
    const float r = gs[j], z = gs[Hd + j], n = gs[2 * Hd + j], hn_lin = gs[3 * Hd + j];
    const float hp = hprev[(long long)d * hprev_ds + (long long)b * Hd + j];
    const float dn_pre = dhv * (1.f - z) * (1.f - n * n);
    const float dz_pre = dhv * (hp - n) * z * (1.f - z);
    const float dr_pre = dn_pre * hn_lin * r * (1.f - r);
    dgir[j] = dr_pre; dgir[Hd + j] = dz_pre; dgir[2 * Hd + j] = dn_pre;
    dghr[j] = dr_pre; dghr[Hd + j] = dz_pre; dghr[2 * Hd + j] = dn_pre * r;
    dh[i] = dhv * z;
}

}  // namespace trid

using namespace trid;

extern "C" int trid_attnpool_tokens_f32(const float* x, const float* pos, float* tok, int B, int T, int C, int ldt,
                                        void* stream) {
    TRID_REQUIRE(x && pos && tok && B > 0 && T > 0 && C > 0 && C % 4 == 0 && ldt >= T + 1, "trid_attnpool_tokens_f32: bad arguments");
    const long long n = (long long)B * (C / 4);
    hipLaunchKernelGGL(attnpool_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)pos, (float4*)tok, B, T, C / 4, ldt);
    return check_launch("trid_attnpool_tokens_f32");
}

extern "C" int trid_attnpool_tokens_fmt_f32(const void* x, int x_fmt, const float* x_amax, const float* pos, float* tok, int B, int T,
                                            int C, int ldt, void* stream) {
    TRID_REQUIRE(x && pos && tok && B > 0 && T > 0 && C > 0 && ldt >= T + 1 && x_fmt >= 0 && x_fmt <= 2, "trid_attnpool_tokens_fmt_f32: bad arguments");
    TRID_REQUIRE(C % (x_fmt == 0 ? 4 : 32) == 0 && (x_fmt != 1 || x_amax), "trid_attnpool_tokens_fmt_f32: C %% 4 (fp32) / C %% 32 (P16, bf16); P16 needs its amax scalar");
    const int CQ = C / 4;
    const dim3 grid((unsigned)((CQ + 63) / 64), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (x_fmt == 1) hipLaunchKernelGGL(attnpool_tokens_wide_kernel<1>, grid, dim3(256), 0, s, (const float4*)x, x_amax, (const float4*)pos, (float4*)tok, T, CQ, ldt);
    else if (x_fmt == 2) hipLaunchKernelGGL(attnpool_tokens_wide_kernel<2>, grid, dim3(256), 0, s, (const float4*)x, x_amax, (const float4*)pos, (float4*)tok, T, CQ, ldt);
    else hipLaunchKernelGGL(attnpool_tokens_wide_kernel<0>, grid, dim3(256), 0, s, (const float4*)x, x_amax, (const float4*)pos, (float4*)tok, T, CQ, ldt);
    return check_launch("trid_attnpool_tokens_fmt_f32");
}

extern "C" int trid_attnpool_tokens_bwd_f32(const float* dtok, float* dx, float* dpos, int B, int T, int C, int ldt,
                                            void* stream) {
    TRID_REQUIRE(dtok && dx && dpos && B > 0 && T > 0 && C % 4 == 0 && ldt >= T + 1, "trid_attnpool_tokens_bwd_f32: bad arguments");
    const long long total = (long long)B * T * (C / 4);
    hipLaunchKernelGGL(attnpool_tokens_bwd_dx_kernel, dim3(grid_for(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dtok, (float4*)dx, B, T, C / 4, ldt);
    const long long n = (long long)(T + 1) * (C / 4);
    hipLaunchKernelGGL(attnpool_tokens_bwd_dpos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)dtok, (float4*)dpos, B, T + 1, C / 4, ldt);
    return check_launch("trid_attnpool_tokens_bwd_f32");
}

extern "C" int trid_softmax_rows_f32(float* s, long long rows, int n, int ld, void* stream) {
    TRID_REQUIRE(s && rows > 0 && n > 0 && ld >= n, "trid_softmax_rows_f32: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, rows,
                       n, ld);
    return check_launch("trid_softmax_rows_f32");
}

extern "C" int trid_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, long long rows, int n, int ld,
                                         void* stream) {
    TRID_REQUIRE(p && dp && ds && rows > 0 && n > 0 && ld >= n, "trid_softmax_rows_bwd_f32: bad arguments");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p,
                       dp, ds, rows, n, ld);
    return check_launch("trid_softmax_rows_bwd_f32");
}

extern "C" int trid_colsum_f32(const float* x, float* out, long long M, int N, long long ld, int accumulate,
                               void* stream) {
    TRID_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "trid_colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, out, M, N, ld,
                       accumulate);
    return check_launch("trid_colsum_f32");
}

extern "C" int trid_embedding_gather_f32(const float* table, const int64_t* tokens, float* x, int B, int L, int ldtok,
                                         int E, long long vocab, void* stream) {
    TRID_REQUIRE(table && tokens && x && B > 0 && L > 0 && ldtok >= L && E % 4 == 0 && vocab > 0,
                 "trid_embedding_gather_f32: bad arguments");
    const long long total = (long long)B * L * (E / 4);
    hipLaunchKernelGGL(embedding_gather_kernel, dim3(grid_for(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)table, tokens, (float4*)x, B, L, ldtok, E / 4, vocab);
    return check_launch("trid_embedding_gather_f32");
}

extern "C" int trid_embedding_bwd_f32(const float* dX, const int64_t* tokens, int B, int L, int ldtok, int E, float* dtable,
                                      long long vocab, long long padding_idx, void* stream) {
    TRID_REQUIRE(dX && tokens && dtable && B > 0 && L > 0 && ldtok >= L && E > 0 && vocab > 0, "trid_embedding_bwd_f32: bad arguments");
    hipError_t e = hipMemsetAsync(dtable, 0, (size_t)vocab * E * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) { set_error("trid_embedding_bwd_f32: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, dX, tokens, ldtok, L, B * L, E, dtable,
                       vocab, padding_idx);
    return check_launch("trid_embedding_bwd_f32");
}

extern "C" int trid_gru_max_init_f32(float* maxv, int32_t* argt, const int64_t* lengths, int Lmax, const int64_t* lmax_dev,
                                     int B, int Hd, void* stream) {
    TRID_REQUIRE(maxv && argt && lengths && B > 0 && Hd > 0 && Lmax > 0, "trid_gru_max_init_f32: bad arguments");
    const int n = B * 2 * Hd;
    hipLaunchKernelGGL(gru_max_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, maxv, argt, lengths,
                       Lmax, lmax_dev, B, Hd);
    return check_launch("trid_gru_max_init_f32");
}

extern "C" int trid_gru_cell_fwd_f32(const float* gi, const float* gh, float* h, const int64_t* lengths, float* gates,
                                     float* hprev, float* maxv, int32_t* argt, int s, int Lmax, int L, int B, int Hd,
                                     long long gates_dstride, long long hprev_dstride, void* stream) {
    TRID_REQUIRE(gi && gh && h && lengths && maxv && argt, "trid_gru_cell_fwd_f32: null pointer");
    TRID_REQUIRE(s >= 0 && s < Lmax && Lmax <= L && B > 0 && Hd > 0, "trid_gru_cell_fwd_f32: bad step/shape");
    const int n = 2 * B * Hd;
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gi, gh, h, lengths,
                       gates, hprev, maxv, argt, s, Lmax, L, B, Hd, gates_dstride, hprev_dstride);
    return check_launch("trid_gru_cell_fwd_f32");
}

extern "C" int trid_gru_cell_bwd_f32(const float* dout, const int32_t* argt, const float* gates, const float* hprev,
                                     const int64_t* lengths, float* dh, float* dGi, float* dgh, int s, int Lmax, int L,
                                     int B, int Hd, long long gates_dstride, long long hprev_dstride,
                                     long long dgh_dstride, void* stream) {
    TRID_REQUIRE(dout && argt && gates && hprev && lengths && dh && dGi && dgh, "trid_gru_cell_bwd_f32: null pointer");
    TRID_REQUIRE(s >= 0 && s < Lmax && Lmax <= L && B > 0 && Hd > 0, "trid_gru_cell_bwd_f32: bad step/shape");
    const int n = 2 * B * Hd;
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dout, argt, gates,
                       hprev, lengths, dh, dGi, dgh, s, Lmax, L, B, Hd, gates_dstride, hprev_dstride, dgh_dstride);
    return check_launch("trid_gru_cell_bwd_f32");
}
