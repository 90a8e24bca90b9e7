// BatchNorm2d (train / eval), ReLU, residual add, AvgPool2d(2), their backward
// passes, the dgrad weight transform and the stem im2col.  All HBM-bound
// streaming kernels over NHWC fp32: 16-byte accesses, grid-stride, channel
// index = element index mod C so loads are fully coalesced.
// Reference: m_resnet.py:19-29,38-49,57-66,161-171 (+ their autograd).

#include "split_common.h"

namespace trid {

// (P16 / bf16 element access: p16_store4, p16_load4, bf16_store4, bf16_load4 - split_common.h)
// scale of a P16 tensor whose largest magnitude is bounded by *a (+ *b): the bound is also published for the consumers
__device__ __forceinline__ float p16_out_scale(const float* a, const float* b, float* sum_out) {
    const float bound = (a != nullptr ? *a : 0.f) + (b != nullptr ? *b : 0.f);
    if (sum_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *sum_out = bound;
    return f16_scale_of(bound);
}

// ---------------------------------------------------------------- weight transform
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int T, int C,
                                        int flip) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int tt = flip ? (T - 1 - t) : t;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, c = c0 + tx;
        tile[i][tx] = (n < N && c < C) ? w[((long long)n * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, n = n0 + tx;
        if (c < C && n < N) wt[((long long)c * T + tt) * N + n] = tile[tx][i];
    }
}

// ---------------------------------------------------------------- stem im2col
__global__ void stem_im2col_kernel(const float* __restrict__ img, float* __restrict__ col, int B, int Cin, int H,
                                   int W, int Ho, int Wo, int ldcol, long long total) {
    const int kk = Cin * 9;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long m = idx / ldcol;
        const int j = (int)(idx - m * ldcol);
        float v = 0.f;
        if (j < kk) {
            const int c = j / 9, r = j - c * 9, ky = r / 3, kx = r - ky * 3;
            const int xo = (int)(m % Wo);
            const long long q = m / Wo;
            const int yo = (int)(q % Ho);
            const int b = (int)(q / Ho);
            const int y = 2 * yo - 1 + ky, x = 2 * xo - 1 + kx;
            if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((long long)b * Cin + c) * H + y) * W + x];
        }
        col[idx] = v;
    }
}

// ---------------------------------------------------------------- BN finalize
// The per-tile (n_b, mean_b, M2_b) partials of a channel (n_b = rows_per_part, the last part ragged) are combined as
//     mean = K + S1 / n,   M2 = sum_b M2_b + S2 - S1^2 / n,   S1 = sum_b n_b (mean_b - K),  S2 = sum_b n_b (mean_b - K)^2
// in fp64 with the shift K = the channel's mean over part 0: only MEANS are ever subtracted and the shifted sums are of the
// size of the spread of the partial means (an unshifted E[x^2] - mean^2 form would amplify the fp32 rounding of the
// partial means by (mean / std)^2: test_bn_finalize_large_mean_offset) - and the sums are plain additions: no division
// per part, any grouping.  pw: floats per (part, channel): 2 = (mean, M2), 4 = (mean, M2, min, max).  With the extremes
// the largest magnitude of act(y * scale + shift) over the channel is known exactly before the apply pass (an affine map
// takes extremes to extremes): folded over the channels into amax_out (zeroed by the caller; integer atomicMax on the
// bit pattern of a non-negative float: order-independent).
// A workgroup owns FIN_CH = 16 adjacent channels (a contiguous 128 / 256 bytes of every part) and one of S ranges of the
// parts; its 256 threads are 16 channels x 16 part slots, each with a handful of independent loads in flight.
//   * up to FIN_ONE_LAUNCH parts: S = 1, the workgroup finalizes its channels itself.
//   * the layers with thousands of parts (stem, layer1: M = 4e5..1.6e6 rows): the range sums go to `scratch` and a second
//     small launch (same thread layout over the S ranges) adds them in range order.  (A single launch with a
//     last-arriver ticket was measured SLOWER: its device-scope release / acquire fences write back and invalidate the
//     whole L2 of an XCD under the GEMMs running beside it.)
struct BnFinalizeOut {
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float momentum, eps;
    float* mean;
    float* invstd;
    float* scale;
    float* shift;
    int relu;
    float* amax_out;  // null: no bound
};

struct FinSums {  // shifted sums of a channel over some of its parts
    double s1, s2, q;
    float lo, hi;
};

constexpr int FIN_CH = 16, FIN_SLOTS = 256 / FIN_CH;

__device__ __forceinline__ void bn_finalize_channel(const BnFinalizeOut& f, int c, double nt, double K, const FinSums& t) {
    const double mu = K + t.s1 / nt;
    double m2 = t.q + t.s2 - t.s1 * t.s1 / nt;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / nt;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float sc = f.gamma[c] * invstd;
    const float shf = f.beta[c] - (float)mu * sc;
    f.mean[c] = (float)mu;
    f.invstd[c] = invstd;
    f.scale[c] = sc;
    f.shift[c] = shf;
    if (f.amax_out != nullptr) {
        const float zl = fmaf(t.lo, sc, shf), zh = fmaf(t.hi, sc, shf);  // the apply pass's own arithmetic
        const float bound = f.relu ? fmaxf(fmaxf(zl, zh), 0.f) : fmaxf(fabsf(zl), fabsf(zh));
        atomicMax(reinterpret_cast<unsigned*>(f.amax_out), __builtin_bit_cast(unsigned, bound));
    }
    if (f.running_mean != nullptr) {
        const double unb = nt > 1.0 ? m2 / (nt - 1.0) : var;
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mu;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
    }
}

// the 16 slots of a channel -> thread (slot 0, channel): fixed order, so the result does not depend on the schedule
__device__ __forceinline__ FinSums fin_fold_slots(FinSums t, double (*sd)[256], float (*sf)[256]) {
    const int tid = threadIdx.x;
    sd[0][tid] = t.s1; sd[1][tid] = t.s2; sd[2][tid] = t.q;
    sf[0][tid] = t.lo; sf[1][tid] = t.hi;
    __syncthreads();
    if (tid < FIN_CH) {
#pragma unroll
        for (int k = 1; k < FIN_SLOTS; ++k) {
            const int j = tid + k * FIN_CH;
            t.s1 += sd[0][j]; t.s2 += sd[1][j]; t.q += sd[2][j];
            t.lo = fminf(t.lo, sf[0][j]); t.hi = fmaxf(t.hi, sf[1][j]);
        }
    }
    return t;
}

// grid: ceil(C / 16) * S workgroups; S == 1: finalizes; S > 1: range sums to scratch [C groups][S][16]
template <int PW>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, int rows_per_part,
                                                          long long M, int C, int S, FinSums* __restrict__ scratch,
                                                          BnFinalizeOut f) {
    __shared__ double sd[3][256];
    __shared__ float sf[2][256];
    const int G = (C + FIN_CH - 1) / FIN_CH;
    const int o = blockIdx.x % G, r = blockIdx.x / G;
    const int tid = threadIdx.x, cl = tid % FIN_CH, slot = tid / FIN_CH;
    const int c = o * FIN_CH + cl;
    const int per = (nparts + S - 1) / S;
    const int begin = r * per, end = min(nparts, begin + per);
    FinSums t{0.0, 0.0, 0.0, INFINITY, -INFINITY};
    double K = 0.0;
    if (c < C) {
        K = (double)partials[(long long)c * PW];
        const double n_full = (double)rows_per_part, n_last = (double)(M - (long long)(nparts - 1) * rows_per_part);
        constexpr int U = 4;  // independent loads in flight per thread
        for (int p0 = begin + slot; p0 < end; p0 += U * FIN_SLOTS) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * FIN_SLOTS;
                if (p < end) {
                    const float* e = partials + ((long long)p * C + c) * PW;
                    if (PW == 4) {
                        v[u] = *reinterpret_cast<const float4*>(e);
                    } else {
                        const float2 w = *reinterpret_cast<const float2*>(e);
                        v[u] = make_float4(w.x, w.y, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * FIN_SLOTS;
                if (p < end) {
                    const double nb = p == nparts - 1 ? n_last : n_full;
                    const double d = (double)v[u].x - K;
                    const double nd = nb * d;
                    t.s1 += nd;
                    t.s2 = fma(nd, d, t.s2);
                    t.q += (double)v[u].y;
                    if (PW == 4) {
                        t.lo = fminf(t.lo, v[u].z);
                        t.hi = fmaxf(t.hi, v[u].w);
                    }
                }
            }
        }
    }
    t = fin_fold_slots(t, sd, sf);
    if (tid < FIN_CH && c < C) {
        if (S == 1) bn_finalize_channel(f, c, (double)M, K, t);
        else scratch[((long long)o * S + r) * FIN_CH + tid] = t;
    }
}

// grid: ceil(C / 16) workgroups: the S range sums of its 16 channels, added in range order
template <int PW>
__global__ __launch_bounds__(256) void bn_finalize_merge_kernel(const float* __restrict__ partials, long long M, int C, int S,
                                                                const FinSums* __restrict__ scratch, BnFinalizeOut f) {
    __shared__ double sd[3][256];
    __shared__ float sf[2][256];
    const int o = blockIdx.x, tid = threadIdx.x, cl = tid % FIN_CH, slot = tid / FIN_CH;
    const int c = o * FIN_CH + cl;
    FinSums t{0.0, 0.0, 0.0, INFINITY, -INFINITY};
    if (c < C) {
        for (int r = slot; r < S; r += FIN_SLOTS) {
            const FinSums e = scratch[((long long)o * S + r) * FIN_CH + cl];
            t.s1 += e.s1; t.s2 += e.s2; t.q += e.q;
            t.lo = fminf(t.lo, e.lo); t.hi = fmaxf(t.hi, e.hi);
        }
    }
    t = fin_fold_slots(t, sd, sf);
    if (tid < FIN_CH && c < C) bn_finalize_channel(f, c, (double)M, (double)partials[(long long)c * PW], t);
}

constexpr int FIN_ONE_LAUNCH = 768;  // parts up to which one workgroup per 16 channels sweeps them all (<= 48 per thread)
constexpr int FIN_MAX_RANGES = 64;
constexpr int FIN_MAX_WG = 2048;  // channel groups x ranges of the two-launch form
constexpr size_t FIN_WS_BYTES = (size_t)FIN_MAX_WG * FIN_CH * sizeof(FinSums);

template <int PW>
static int bn_finalize_launch(const float* partials, int nparts, int rows_per_part, long long M, int C, void* ws,
                              const BnFinalizeOut& f, hipStream_t stream) {
    const int G = (C + FIN_CH - 1) / FIN_CH;
    if (ws != nullptr && nparts > FIN_ONE_LAUNCH && 2 * G <= FIN_MAX_WG) {
        int S = (nparts + 12 * FIN_SLOTS - 1) / (12 * FIN_SLOTS);  // ~12 parts per thread
        if (S > FIN_MAX_RANGES) S = FIN_MAX_RANGES;
        if (S > FIN_MAX_WG / G) S = FIN_MAX_WG / G;
        FinSums* scratch = reinterpret_cast<FinSums*>(ws);
        hipLaunchKernelGGL(bn_finalize_kernel<PW>, dim3(G * S), dim3(256), 0, stream, partials, nparts, rows_per_part, M, C, S,
                           scratch, f);
        hipLaunchKernelGGL(bn_finalize_merge_kernel<PW>, dim3(G), dim3(256), 0, stream, partials, M, C, S,
                           (const FinSums*)scratch, f);
        return 0;
    }
    hipLaunchKernelGGL(bn_finalize_kernel<PW>, dim3(G), dim3(256), 0, stream, partials, nparts, rows_per_part, M, C, 1,
                       (FinSums*)nullptr, f);
    return 0;
}

// eval mode: max|act(y * scale + shift)| of a conv output from the (mean, M2, min, max) partials of its epilogue and the
// running-statistics coefficients -> atomicMax into *bound (zeroed by the caller).  An affine map takes the extremes of a
// set to extremes, and the largest act(.) over a channel is reached at the (min or max) of one of its parts - so every
// (part, channel) entry is processed on its own and only a global maximum is folded: no per-channel reduction.
__global__ __launch_bounds__(256) void bn_eval_bound_kernel(const float4* __restrict__ partials, long long total, int C,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int relu, float* __restrict__ bound) {
    float m = 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const float4 v = partials[e];
        const float zl = fmaf(v.z, scale[c], shift[c]), zh = fmaf(v.w, scale[c], shift[c]);
        m = fmaxf(m, relu ? fmaxf(zl, zh) : fmaxf(fabsf(zl), fabsf(zh)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(bound), __builtin_bit_cast(unsigned, m));
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float invstd = 1.f / sqrtf(rv[c] + eps);
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// ---------------------------------------------------------------- amax side output
// The kernels that PRODUCE a GEMM operand can also fold max|out| into a device scalar (the operand scale of the
// fp16-split GEMM arithmetic, trid_gemm_desc.precision == 16): per-thread running maximum, one wave / block fold,
// one fire-and-forget integer atomicMax on the bit pattern per block.  `amax` must start at 0.
__device__ __forceinline__ unsigned amax4(unsigned m, float4 v) {
    const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
    const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    const unsigned q = ab > cd ? ab : cd;
    return q > m ? q : m;
}
__device__ __forceinline__ void amax_commit(unsigned m, float* amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
    }
    __shared__ unsigned amax_red[8];
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) amax_red[w] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = 0;
        for (int i = 0; i < nw; ++i) r = amax_red[i] > r ? amax_red[i] : r;
        if (r != 0) atomicMax(reinterpret_cast<unsigned*>(amax), r);
    }
}

// ---------------------------------------------------------------- BN apply (+res, +relu)
__device__ __forceinline__ float4 affine4(float4 v, float4 s, float4 t) {
    return make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
}
__device__ __forceinline__ float4 relu4(float4 v) {
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

// P16OUT: `out` is written as a P16 tensor whose scale comes from the bound *oa (+ *ob), known BEFORE this pass
// (bn_finalize's extremes; a residual adds its own bound); the sum is published in *osum for the consumers.
// P16RES: the identity residual `res` is a P16 tensor (scale from *res_amax): decoded on the fly.
// (formats: 0 = fp32, 1 = P16, 2 = plain bf16)
// (32 VGPRs: next to the 3x3 tile GEMM - four waves of 120 VGPRs per SIMD - one wave of this kernel still fits a SIMD's
// register file, so the key encoder's apply passes make some progress under the query encoder's convolutions; measured
// 44.12 -> 44.02 ms per step, no scratch)
template <int OFMT, int RFMT>
__global__ __attribute__((amdgpu_num_vgpr(32))) void bn_apply_kernel(const float4* __restrict__ y, const float4* __restrict__ scale,
                                const float4* __restrict__ shift, const float4* __restrict__ res,
                                const float4* __restrict__ rscale, const float4* __restrict__ rshift,
                                float4* __restrict__ out, long long total4, int CQ, int relu,
                                unsigned long long* __restrict__ mask, float* __restrict__ amax,
                                const float* __restrict__ oa, const float* __restrict__ ob, float* __restrict__ osum,
                                const float* __restrict__ res_amax, int y_fmt, int nt) {
    unsigned am = 0;
    constexpr bool P16OUT = OFMT == 1;
    const float oscale = P16OUT ? p16_out_scale(oa, ob, osum) : 1.f;
    const float rinv = RFMT == 1 ? 1.f / f16_scale_of(*res_amax) : 1.f;
    // the (row, channel quad) of element i is carried along - one add and a conditional subtract per trip - instead of being
    // divided out of i per element: i % CQ here, i / CQ inside the P16 load and the P16 store were three 64-bit divisions per
    // element, ~25 VALU instructions and two exec-mask branches each, next to ~30 instructions of real work
    const long long T = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int row = (int)(i0 / CQ);
    int cq = (int)(i0 - (long long)row * CQ);
    const int step_r = (int)(T / CQ), step_c = (int)(T - (long long)step_r * CQ);
    for (long long i = i0; i < total4; i += T, cq += step_c, row += step_r) {
        if (cq >= CQ) { cq -= CQ; ++row; }
        // y_fmt 2: the raw conv output itself is a bf16 tensor (bf16 mode)
        float4 v = affine4(y_fmt == 2 ? bf16_load4(reinterpret_cast<const uint2*>(y), i) : ld_stream4(y + i, nt), scale[cq], shift[cq]);
        if (res != nullptr) {
            float4 r = RFMT == 1 ? p16_load4_rc(reinterpret_cast<const uint2*>(res), row, cq, CQ, rinv)
                     : RFMT == 2 ? bf16_load4(reinterpret_cast<const uint2*>(res), i) : ld_stream4(res + i, nt);
            if (rscale != nullptr) r = affine4(r, rscale[cq], rshift[cq]);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (mask != nullptr) {
            // 1 bit per element for the backward pass (instead of re-reading the output twice):
            // quad i -> words (i/64)*4 + c, bit i%64; a wave covers 64 consecutive quads
            const unsigned long long bx = __ballot(v.x > 0.f), by = __ballot(v.y > 0.f);
            const unsigned long long bz = __ballot(v.z > 0.f), bw = __ballot(v.w > 0.f);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long* m = mask + (i >> 6) * 4;
                m[0] = bx; m[1] = by; m[2] = bz; m[3] = bw;
            }
        }
        if (relu) v = relu4(v);
        if (OFMT == 1) {
            p16_store4_pair_rc(reinterpret_cast<uint2*>(out), row, cq, CQ, v, oscale, nt);  // (total4 is even and the stride a multiple of 256: lane pairs hold quad pairs)
        } else if (OFMT == 2) {
            bf16_store4(reinterpret_cast<uint2*>(out), i, v);
        } else {
            st_stream4(out + i, v, nt);
            am = amax4(am, v);
        }
    }
    if (OFMT == 0 && amax != nullptr) amax_commit(am, amax);
}

// P16OUT as above (scale from *oa); P16IN: `y` is a P16 tensor (scale from *in_amax) - the plain pooling of a block
// input on its way to the downsample convolution keeps the input's scale (an average never exceeds the maximum)
template <int OFMT, int IFMT>
__global__ void bn_apply_pool2_kernel(const float4* __restrict__ y, const float4* __restrict__ scale,
                                      const float4* __restrict__ shift, float4* __restrict__ out, int B, int H, int W,
                                      int CQ, int relu, long long total4, float* __restrict__ amax,
                                      const float* __restrict__ oa, const float* __restrict__ in_amax) {
    const int Ho = H / 2, Wo = W / 2;
    unsigned am = 0;
    const float oscale = OFMT == 1 ? f16_scale_of(*oa) : 1.f;
    const float iinv = IFMT == 1 ? 1.f / f16_scale_of(*in_amax) : 1.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % CQ);
        long long p = i / CQ;
        const int xo = (int)(p % Wo);
        p /= Wo;
        const int yo = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale != nullptr) { s = scale[cq]; t = shift[cq]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long long src = (((long long)b * H + 2 * yo + dy) * W + 2 * xo + dx) * CQ + cq;
                float4 v = IFMT == 1 ? p16_load4(reinterpret_cast<const uint2*>(y), src, CQ, iinv)
                         : IFMT == 2 ? bf16_load4(reinterpret_cast<const uint2*>(y), src) : y[src];
                v = affine4(v, s, t);
                if (relu) v = relu4(v);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        const float4 o = make_float4(acc.x * 0.25f, acc.y * 0.25f, acc.z * 0.25f, acc.w * 0.25f);
        if (OFMT == 1) {
            p16_store4_pair(reinterpret_cast<uint2*>(out), i, CQ, o, oscale);
        } else if (OFMT == 2) {
            bf16_store4(reinterpret_cast<uint2*>(out), i, o);
        } else {
            out[i] = o;
            am = amax4(am, o);
        }
    }
    if (OFMT == 0 && amax != nullptr) amax_commit(am, amax);
}

// FMT 0: fp32 tensors; 2: plain bf16 (gradient tensors of configs[3]'s bf16 mode; 0.25 * a bf16 value is exact)
template <int FMT>
__global__ void avgpool2_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dx, int B, int H, int W, int CQ,
                                    int accumulate, long long total4) {
    const int Ho = H / 2, Wo = W / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % CQ);
        long long p = i / CQ;
        const int x = (int)(p % W);
        p /= W;
        const int yy = (int)(p % H);
        const int b = (int)(p / H);
        const long long src = (((long long)b * Ho + yy / 2) * Wo + x / 2) * CQ + cq;
        float4 v = FMT == 2 ? bf16_load4(reinterpret_cast<const uint2*>(g), src) : g[src];
        v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f;
        if (accumulate) {
            const float4 o = FMT == 2 ? bf16_load4(reinterpret_cast<const uint2*>(dx), i) : dx[i];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        if (FMT == 2) bf16_store4(reinterpret_cast<uint2*>(dx), i, v);
        else dx[i] = v;
    }
}

// ---------------------------------------------------------------- BN backward
struct BnBwdArgs {
    const float4* g;
    const float4* y;
    const float4* act;
    const float4* mean;
    const float4* invstd;
    const float4* scale;
    const float4* shift;
    int mask_mode, pooled;
    int g_fmt;  // 0: g (and dres) fp32; 2: plain bf16 (configs[3]: gradients of bf16 tensors are bf16 tensors)
    int y_fmt;  // 0: y fp32; 2: plain bf16 (configs[3]: the conv outputs themselves)
    int B, H, W, CQ;
    long long total4;  // B*H*W*CQ
    FastDiv fdW, fdH;
    int nt;            // non-temporal streaming accesses (tensors beyond the Infinity Cache: split_common.h)
};

// masked effective gradient and xhat for element i (channel quad cq)
__device__ __forceinline__ void bn_bwd_elem(const BnBwdArgs& a, long long i, int cq, float4& gm, float4& xh) {
    float4 g;
    if (a.pooled) {
        const uint32_t p = (uint32_t)(i / a.CQ);
        const uint32_t q = fdiv(p, a.fdW);
        const int x = (int)(p - q * a.W);
        const uint32_t b = fdiv(q, a.fdH);
        const int yy = (int)(q - b * a.H);
        const long long src = (((long long)b * (a.H / 2) + yy / 2) * (a.W / 2) + x / 2) * a.CQ + cq;
        g = a.g_fmt == 2 ? bf16_load4(reinterpret_cast<const uint2*>(a.g), src) : a.g[src];
        g.x *= 0.25f; g.y *= 0.25f; g.z *= 0.25f; g.w *= 0.25f;
    } else {
        g = a.g_fmt == 2 ? bf16_load4(reinterpret_cast<const uint2*>(a.g), i) : ld_stream4(a.g + i, a.nt);
    }
    const float4 yv = a.y_fmt == 2 ? bf16_load4(reinterpret_cast<const uint2*>(a.y), i) : ld_stream4(a.y + i, a.nt);
    const float4 mu = a.mean[cq], is = a.invstd[cq];
    xh = make_float4((yv.x - mu.x) * is.x, (yv.y - mu.y) * is.y, (yv.z - mu.z) * is.z, (yv.w - mu.w) * is.w);
    if (a.mask_mode == 1) {
        const float4 z = affine4(yv, a.scale[cq], a.shift[cq]);
        g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f;
        g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
    } else if (a.mask_mode == 2) {
        const float4 z = a.act[i];
        g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f;
        g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
    } else if (a.mask_mode == 3) {  // bit mask written by bn_apply
        const unsigned long long* m = reinterpret_cast<const unsigned long long*>(a.act) + (i >> 6) * 4;
        const int bit = (int)(i & 63);
        g.x = ((m[0] >> bit) & 1ull) ? g.x : 0.f; g.y = ((m[1] >> bit) & 1ull) ? g.y : 0.f;
        g.z = ((m[2] >> bit) & 1ull) ? g.z : 0.f; g.w = ((m[3] >> bit) & 1ull) ? g.w : 0.f;
    }
    gm = g;
}

// grid*256 is a multiple of CQ, so a thread always sees the same channel quad.
// ws layout: [grid][CW][8] with CW = min(CQ,256) quads covered by a block.
// BOUND: also the per-channel maxima of |masked g| and |xhat| (ws2 [grid][CW][8]) - with dgamma / dbeta they bound
// |dy| BEFORE the apply pass, which then writes dy as a P16 tensor (bn_bwd_reduce_final_kernel).
template <bool BOUND>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a, float* __restrict__ ws, float* __restrict__ ws2) {
    const int tid = threadIdx.x;
    const long long T = (long long)gridDim.x * 256;
    const long long t0 = (long long)blockIdx.x * 256 + tid;
    const int cq = (int)(t0 % a.CQ);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mg = s1, mx = s1;
    auto fold = [&](const float4& gm, const float4& xh) {
        s1.x += gm.x; s1.y += gm.y; s1.z += gm.z; s1.w += gm.w;
        s2.x = fmaf(gm.x, xh.x, s2.x); s2.y = fmaf(gm.y, xh.y, s2.y);
        s2.z = fmaf(gm.z, xh.z, s2.z); s2.w = fmaf(gm.w, xh.w, s2.w);
        if (BOUND) {
            mg.x = fmaxf(mg.x, fabsf(gm.x)); mg.y = fmaxf(mg.y, fabsf(gm.y)); mg.z = fmaxf(mg.z, fabsf(gm.z)); mg.w = fmaxf(mg.w, fabsf(gm.w));
            mx.x = fmaxf(mx.x, fabsf(xh.x)); mx.y = fmaxf(mx.y, fabsf(xh.y)); mx.z = fmaxf(mx.z, fabsf(xh.z)); mx.w = fmaxf(mx.w, fabsf(xh.w));
        }
    };
    // (measured and dropped, profiles/r06k_bn_ilp_ab.txt: four elements of a thread in flight - loads issued back to back, folded in
    // this order - run the isolated pass 1-2 % faster but need 126 VGPRs instead of 64, and the step 0.25 ms SLOWER: these passes
    // share the CUs with the other lanes' GEMMs, whose waves leave room for a small kernel's waves only)
    for (long long i = t0; i < a.total4; i += T) {
        float4 gm, xh;
        bn_bwd_elem(a, i, cq, gm, xh);
        fold(gm, xh);
    }
    __shared__ float red[256][9];
    red[tid][0] = s1.x; red[tid][1] = s1.y; red[tid][2] = s1.z; red[tid][3] = s1.w;
    red[tid][4] = s2.x; red[tid][5] = s2.y; red[tid][6] = s2.z; red[tid][7] = s2.w;
    __syncthreads();
    const int CW = a.CQ < 256 ? a.CQ : 256;
    if (tid < CW) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int r = tid; r < 256; r += CW)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += red[r][k];
        float* dst = ws + bn_bwd_partial_index(blockIdx.x, tid, gridDim.x, a.CQ);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = acc[k];
    }
    if (BOUND) {
        __syncthreads();
        red[tid][0] = mg.x; red[tid][1] = mg.y; red[tid][2] = mg.z; red[tid][3] = mg.w;
        red[tid][4] = mx.x; red[tid][5] = mx.y; red[tid][6] = mx.z; red[tid][7] = mx.w;
        __syncthreads();
        if (tid < CW) {
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.f;
            for (int r = tid; r < 256; r += CW)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaxf(acc[k], red[r][k]);
            float* dst = ws2 + bn_bwd_partial_index(blockIdx.x, tid, gridDim.x, a.CQ);
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = acc[k];
        }
    }
}

// one workgroup per channel quad: sum the block partials that cover it
// ws2 / scale / bound (all or none): |dy_c| <= |scale_c| * (max|g_c| + (|dbeta_c| + max|xhat_c| * |dgamma_c|) / M), folded
// over the channels into *bound (zeroed by the caller; integer atomicMax on a non-negative float's bit pattern)
__global__ __launch_bounds__(256) void bn_bwd_reduce_final_kernel(const float* __restrict__ ws, int nblk, int CQ,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  int C, const float* __restrict__ ws2,
                                                                  const float* __restrict__ scale, float invM,
                                                                  float* __restrict__ bound) {
    const int q = blockIdx.x;
    const int CW = CQ < 256 ? CQ : 256;
    const int S = CQ / CW;  // channel-quad slices; block b covers slice b % S
    const int slice = q / CW, ql = q % CW;
    float acc[8], mxv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = mxv[k] = 0.f;
    for (int b = slice + S * threadIdx.x; b < nblk; b += S * 256) {
        const float4* src = reinterpret_cast<const float4*>(ws + bn_bwd_partial_index(b, ql, nblk, CQ));
        const float4 u = src[0], v = src[1];
        acc[0] += u.x; acc[1] += u.y; acc[2] += u.z; acc[3] += u.w;
        acc[4] += v.x; acc[5] += v.y; acc[6] += v.z; acc[7] += v.w;
        if (ws2 != nullptr) {
            const float4* s2 = reinterpret_cast<const float4*>(ws2 + bn_bwd_partial_index(b, ql, nblk, CQ));
            const float4 a = s2[0], c = s2[1];
            mxv[0] = fmaxf(mxv[0], a.x); mxv[1] = fmaxf(mxv[1], a.y); mxv[2] = fmaxf(mxv[2], a.z); mxv[3] = fmaxf(mxv[3], a.w);
            mxv[4] = fmaxf(mxv[4], c.x); mxv[5] = fmaxf(mxv[5], c.y); mxv[6] = fmaxf(mxv[6], c.z); mxv[7] = fmaxf(mxv[7], c.w);
        }
    }
    __shared__ float red[4][8], redm[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        acc[k] = wave_sum(acc[k]);
        mxv[k] = wave_max(mxv[k]);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            red[threadIdx.x >> 6][k] = acc[k];
            redm[threadIdx.x >> 6][k] = mxv[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int c = 4 * q + (threadIdx.x & 3);
        if (c < C) {
            if (threadIdx.x < 4) dbeta[c] = s; else dgamma[c] = s;
        }
    }
    if (ws2 != nullptr && threadIdx.x < 4) {
        const int k = threadIdx.x, c = 4 * q + k;
        if (c < C) {
            const float db = red[0][k] + red[1][k] + red[2][k] + red[3][k];
            const float dg = red[0][4 + k] + red[1][4 + k] + red[2][4 + k] + red[3][4 + k];
            const float mg = fmaxf(fmaxf(redm[0][k], redm[1][k]), fmaxf(redm[2][k], redm[3][k]));
            const float mx = fmaxf(fmaxf(redm[0][4 + k], redm[1][4 + k]), fmaxf(redm[2][4 + k], redm[3][4 + k]));
            // 1.0001: the apply pass rounds its three-term expression differently from this bound
            const float b = 1.0001f * fabsf(scale[c]) * (mg + (fabsf(db) + mx * fabsf(dg)) * invM);
            atomicMax(reinterpret_cast<unsigned*>(bound), __builtin_bit_cast(unsigned, b));
        }
    }
}

// P16OUT: dy is written as a P16 tensor, scale from the bound *oa of the reduce pass
template <int OFMT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a, const float4* __restrict__ dgamma,
                                                           const float4* __restrict__ dbeta, float invM,
                                                           float4* __restrict__ dy, float4* __restrict__ dres,
                                                           float* __restrict__ amax, const float* __restrict__ oa) {
    unsigned am = 0;
    const float oscale = OFMT == 1 ? f16_scale_of(*oa) : 1.f;
    const long long T = (long long)gridDim.x * 256;
    auto finish = [&](long long i, long long row, int cq, const float4& gm, const float4& xh) {
        const float4 sc = a.scale[cq], dg = dgamma[cq], db = dbeta[cq];
        float4 o;
        o.x = sc.x * (gm.x - db.x * invM - xh.x * dg.x * invM);
        o.y = sc.y * (gm.y - db.y * invM - xh.y * dg.y * invM);
        o.z = sc.z * (gm.z - db.z * invM - xh.z * dg.z * invM);
        o.w = sc.w * (gm.w - db.w * invM - xh.w * dg.w * invM);
        if (OFMT == 1) {
            p16_store4_pair_rc(reinterpret_cast<uint2*>(dy), row, cq, a.CQ, o, oscale, a.nt);
        } else if (OFMT == 2) {
            bf16_store4(reinterpret_cast<uint2*>(dy), i, o);
        } else {
            st_stream4(dy + i, o, a.nt);
            am = amax4(am, o);
        }
        if (dres != nullptr) {
            if (a.g_fmt == 2) bf16_store4(reinterpret_cast<uint2*>(dres), i, gm);  // (a masked bf16 value: exact)
            else st_stream4(dres + i, gm, a.nt);
        }
    };
    // the (row, channel quad) of element i: carried along (one add and a conditional subtract per trip) instead of a 64-bit i % CQ per
    // element - the remainder's expansion is ~25 VALU instructions and two exec-mask branches
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    long long row = i0 / a.CQ;
    int cq = (int)(i0 - row * a.CQ);
    const long long step_r = T / a.CQ;
    const int step_c = (int)(T - step_r * a.CQ);
    auto advance = [&](long long& r, int& c) {
        c += step_c;
        r += step_r;
        if (c >= a.CQ) { c -= a.CQ; ++r; }
    };
    long long i = i0;
    for (; i < a.total4; i += T) {
        float4 gm, xh;
        bn_bwd_elem(a, i, cq, gm, xh);
        finish(i, row, cq, gm, xh);
        advance(row, cq);
    }
    if (OFMT == 0 && amax != nullptr) amax_commit(am, amax);
}

// ---- the two BatchNorm layers that share a gradient: a downsample block's bn3 and the BatchNorm of its downsample branch
// (m_resnet.py:62-66: out = relu(bn3(conv3(.)) + downsample(x))) both receive g masked by the block's ReLU bits.  One pass
// reads g, the bits and BOTH saved conv outputs: sum g m is shared, sum g m xhat and the maxima are kept per layer, in the
// workspaces bn_bwd_reduce_final_kernel folds (one call per layer).  12 instead of 16 bytes per element for the sums,
// 20 instead of 24 for the apply pass, two launches instead of four.
struct BnBwdDual {
    const float4* g;
    const unsigned long long* bits;  // relu_mask of the block output
    const float4* y1;
    const float4* y2;
    const float4* mean1;
    const float4* invstd1;
    const float4* scale1;
    const float4* mean2;
    const float4* invstd2;
    const float4* scale2;
    int CQ;
    long long total4;
    int nt;
};

__device__ __forceinline__ void bn_bwd_dual_elem(const BnBwdDual& a, long long i, int cq, float4& gm, float4& xh1, float4& xh2) {
    float4 g = ld_stream4(a.g + i, a.nt);
    const float4 ya = ld_stream4(a.y1 + i, a.nt), yb = ld_stream4(a.y2 + i, a.nt);
    const unsigned long long* m = a.bits + (i >> 6) * 4;
    const int bit = (int)(i & 63);
    g.x = ((m[0] >> bit) & 1ull) ? g.x : 0.f; g.y = ((m[1] >> bit) & 1ull) ? g.y : 0.f;
    g.z = ((m[2] >> bit) & 1ull) ? g.z : 0.f; g.w = ((m[3] >> bit) & 1ull) ? g.w : 0.f;
    const float4 mu1 = a.mean1[cq], is1 = a.invstd1[cq], mu2 = a.mean2[cq], is2 = a.invstd2[cq];
    xh1 = make_float4((ya.x - mu1.x) * is1.x, (ya.y - mu1.y) * is1.y, (ya.z - mu1.z) * is1.z, (ya.w - mu1.w) * is1.w);
    xh2 = make_float4((yb.x - mu2.x) * is2.x, (yb.y - mu2.y) * is2.y, (yb.z - mu2.z) * is2.z, (yb.w - mu2.w) * is2.w);
    gm = g;
}

// ws layout per layer as bn_bwd_reduce_kernel<true>: wsA / wsB = [grid][CW][8] sums (s1 | s2), ws2A / ws2B = maxima (|g m| | |xhat|)
__global__ __launch_bounds__(256) void bn_bwd_dual_reduce_kernel(BnBwdDual a, float* __restrict__ wsA, float* __restrict__ ws2A,
                                                                 float* __restrict__ wsB, float* __restrict__ ws2B) {
    const int tid = threadIdx.x;
    const long long T = (long long)gridDim.x * 256;
    const long long t0 = (long long)blockIdx.x * 256 + tid;
    const int cq = (int)(t0 % a.CQ);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), sa = s1, sb = s1, mg = s1, ma = s1, mb = s1;
    for (long long i = t0; i < a.total4; i += T) {
        float4 gm, x1, x2;
        bn_bwd_dual_elem(a, i, cq, gm, x1, x2);
        s1.x += gm.x; s1.y += gm.y; s1.z += gm.z; s1.w += gm.w;
        sa.x = fmaf(gm.x, x1.x, sa.x); sa.y = fmaf(gm.y, x1.y, sa.y); sa.z = fmaf(gm.z, x1.z, sa.z); sa.w = fmaf(gm.w, x1.w, sa.w);
        sb.x = fmaf(gm.x, x2.x, sb.x); sb.y = fmaf(gm.y, x2.y, sb.y); sb.z = fmaf(gm.z, x2.z, sb.z); sb.w = fmaf(gm.w, x2.w, sb.w);
        mg.x = fmaxf(mg.x, fabsf(gm.x)); mg.y = fmaxf(mg.y, fabsf(gm.y)); mg.z = fmaxf(mg.z, fabsf(gm.z)); mg.w = fmaxf(mg.w, fabsf(gm.w));
        ma.x = fmaxf(ma.x, fabsf(x1.x)); ma.y = fmaxf(ma.y, fabsf(x1.y)); ma.z = fmaxf(ma.z, fabsf(x1.z)); ma.w = fmaxf(ma.w, fabsf(x1.w));
        mb.x = fmaxf(mb.x, fabsf(x2.x)); mb.y = fmaxf(mb.y, fabsf(x2.y)); mb.z = fmaxf(mb.z, fabsf(x2.z)); mb.w = fmaxf(mb.w, fabsf(x2.w));
    }
    __shared__ float red[256][13];
    const int CW = a.CQ < 256 ? a.CQ : 256;
    // sums: s1 (shared), sa, sb
    red[tid][0] = s1.x; red[tid][1] = s1.y; red[tid][2] = s1.z; red[tid][3] = s1.w;
    red[tid][4] = sa.x; red[tid][5] = sa.y; red[tid][6] = sa.z; red[tid][7] = sa.w;
    red[tid][8] = sb.x; red[tid][9] = sb.y; red[tid][10] = sb.z; red[tid][11] = sb.w;
    __syncthreads();
    if (tid < CW) {
        float acc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] = 0.f;
        for (int r = tid; r < 256; r += CW)
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k] += red[r][k];
        float* da = wsA + bn_bwd_partial_index(blockIdx.x, tid, gridDim.x, a.CQ);
        float* db = wsB + bn_bwd_partial_index(blockIdx.x, tid, gridDim.x, a.CQ);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            da[k] = acc[k]; db[k] = acc[k];
            da[4 + k] = acc[4 + k]; db[4 + k] = acc[8 + k];
        }
    }
    __syncthreads();
    red[tid][0] = mg.x; red[tid][1] = mg.y; red[tid][2] = mg.z; red[tid][3] = mg.w;
    red[tid][4] = ma.x; red[tid][5] = ma.y; red[tid][6] = ma.z; red[tid][7] = ma.w;
    red[tid][8] = mb.x; red[tid][9] = mb.y; red[tid][10] = mb.z; red[tid][11] = mb.w;
    __syncthreads();
    if (tid < CW) {
        float acc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] = 0.f;
        for (int r = tid; r < 256; r += CW)
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k] = fmaxf(acc[k], red[r][k]);
        float* da = ws2A + bn_bwd_partial_index(blockIdx.x, tid, gridDim.x, a.CQ);
        float* db = ws2B + bn_bwd_partial_index(blockIdx.x, tid, gridDim.x, a.CQ);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            da[k] = acc[k]; db[k] = acc[k];
            da[4 + k] = acc[4 + k]; db[4 + k] = acc[8 + k];
        }
    }
}

// dy1 / dy2: P16 tensors scaled for *bound1 / *bound2 (the folds of the two layers)
__global__ __launch_bounds__(256) void bn_bwd_dual_apply_kernel(BnBwdDual a, const float4* __restrict__ dgamma1, const float4* __restrict__ dgamma2,
                                                                const float4* __restrict__ dbeta, float invM, float4* __restrict__ dy1,
                                                                float4* __restrict__ dy2, const float* __restrict__ bound1,
                                                                const float* __restrict__ bound2) {
    const float os1 = f16_scale_of(*bound1), os2 = f16_scale_of(*bound2);
    // ((row, channel quad) carried along instead of three divisions of i per element: see bn_apply_kernel)
    const long long T = (long long)gridDim.x * 256;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    long long row = i0 / a.CQ;
    int cq = (int)(i0 - row * a.CQ);
    const long long step_r = T / a.CQ;
    const int step_c = (int)(T - step_r * a.CQ);
    for (long long i = i0; i < a.total4; i += T, cq += step_c, row += step_r) {
        if (cq >= a.CQ) { cq -= a.CQ; ++row; }
        float4 gm, x1, x2;
        bn_bwd_dual_elem(a, i, cq, gm, x1, x2);
        const float4 db = dbeta[cq];
        {
            const float4 sc = a.scale1[cq], dg = dgamma1[cq];
            float4 o;
            o.x = sc.x * (gm.x - db.x * invM - x1.x * dg.x * invM);
            o.y = sc.y * (gm.y - db.y * invM - x1.y * dg.y * invM);
            o.z = sc.z * (gm.z - db.z * invM - x1.z * dg.z * invM);
            o.w = sc.w * (gm.w - db.w * invM - x1.w * dg.w * invM);
            p16_store4_pair_rc(reinterpret_cast<uint2*>(dy1), row, cq, a.CQ, o, os1, a.nt);
        }
        {
            const float4 sc = a.scale2[cq], dg = dgamma2[cq];
            float4 o;
            o.x = sc.x * (gm.x - db.x * invM - x2.x * dg.x * invM);
            o.y = sc.y * (gm.y - db.y * invM - x2.y * dg.y * invM);
            o.z = sc.z * (gm.z - db.z * invM - x2.z * dg.z * invM);
            o.w = sc.w * (gm.w - db.w * invM - x2.w * dg.w * invM);
            p16_store4_pair_rc(reinterpret_cast<uint2*>(dy2), row, cq, a.CQ, o, os2, a.nt);
        }
    }
}

// non-temporal accesses for this launch?  (bytes of ONE fp32-sized tensor of the pass; TRID_BN_NT=0 / 1 forces it off / on)
static int stream_nt(long long tensor_bytes) {
    static const int env = getenv("TRID_BN_NT") ? atoi(getenv("TRID_BN_NT")) : -1;
    return env >= 0 ? (env != 0) : (tensor_bytes >= STREAM_NT_MIN_BYTES);
}

static int bn_bwd_grid(long long total4, int CQ) {
    // multiple of S = CQ/min(CQ,256) so that grid*256 % CQ == 0
    const int CW = CQ < 256 ? CQ : 256;
    const int S = CQ / CW;
    long long g = (total4 + 256LL * 8 - 1) / (256LL * 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    g = (g + S - 1) / S * S;
    return (int)g;
}

}  // namespace trid

using namespace trid;

extern "C" int trid_weight_transpose_f32(const float* w, float* wt, int N, int T, int C, int flip, void* stream) {
    TRID_REQUIRE(w && wt && N > 0 && T > 0 && C > 0, "trid_weight_transpose_f32: bad arguments");
    dim3 grid((C + 31) / 32, (N + 31) / 32, T);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, N, T, C, flip);
    return check_launch("trid_weight_transpose_f32");
}

extern "C" int trid_stem_im2col_f32(const float* img, float* col, int B, int Cin, int H, int W, int Ho, int Wo,
                                    int ldcol, void* stream) {
    TRID_REQUIRE(img && col && B > 0 && Cin > 0 && ldcol >= Cin * 9, "trid_stem_im2col_f32: bad arguments");
    TRID_REQUIRE(Ho == (H + 1) / 2 && Wo == (W + 1) / 2, "trid_stem_im2col_f32: Ho/Wo must be ceil(H/2), ceil(W/2)");
    const long long total = (long long)B * Ho * Wo * ldcol;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3(grid_for(total, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, img,
                       col, B, Cin, H, W, Ho, Wo, ldcol, total);
    return check_launch("trid_stem_im2col_f32");
}

extern "C" long long trid_bn_finalize_ws_bytes(void) { return (long long)FIN_WS_BYTES; }

extern "C" int trid_bn_finalize_f32(const float* partials, int nparts, int rows_per_part, long long M, int C,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                                    void* ws, void* stream) {
    TRID_REQUIRE(partials && gamma && beta && mean && invstd && scale && shift, "trid_bn_finalize_f32: null pointer");
    TRID_REQUIRE(aligned16(partials) && aligned16(ws), "trid_bn_finalize_f32: partials / ws must be 16-byte aligned");
    TRID_REQUIRE(nparts > 0 && rows_per_part > 0 && C > 0 && M > (long long)(nparts - 1) * rows_per_part &&
                     M <= (long long)nparts * rows_per_part,
                 "trid_bn_finalize_f32: nparts=%d rows_per_part=%d inconsistent with M=%lld", nparts, rows_per_part, M);
    TRID_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "trid_bn_finalize_f32: running stats both or none");
    const BnFinalizeOut f{gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, 0, nullptr};
    bn_finalize_launch<2>(partials, nparts, rows_per_part, M, C, ws, f, (hipStream_t)stream);
    return check_launch("trid_bn_finalize_f32");
}

extern "C" int trid_bn_finalize_minmax_f32(const float* partials, int nparts, int rows_per_part, long long M, int C,
                                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                                           float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                                           int relu, float* amax_out, void* ws, void* stream) {
    TRID_REQUIRE(partials && gamma && beta && mean && invstd && scale && shift && amax_out, "trid_bn_finalize_minmax_f32: null pointer");
    TRID_REQUIRE(aligned16(partials) && aligned16(ws), "trid_bn_finalize_minmax_f32: partials / ws must be 16-byte aligned");
    TRID_REQUIRE(nparts > 0 && rows_per_part > 0 && C > 0 && M > (long long)(nparts - 1) * rows_per_part &&
                     M <= (long long)nparts * rows_per_part,
                 "trid_bn_finalize_minmax_f32: nparts=%d rows_per_part=%d inconsistent with M=%lld", nparts, rows_per_part, M);
    TRID_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "trid_bn_finalize_minmax_f32: running stats both or none");
    const BnFinalizeOut f{gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, relu, amax_out};
    bn_finalize_launch<4>(partials, nparts, rows_per_part, M, C, ws, f, (hipStream_t)stream);
    return check_launch("trid_bn_finalize_minmax_f32");
}

extern "C" int trid_bn_eval_bound_f32(const float* partials, int nparts, int C, const float* scale, const float* shift, int relu,
                                      float* bound, void* stream) {
    TRID_REQUIRE(partials && scale && shift && bound && nparts > 0 && C > 0 && aligned16(partials), "trid_bn_eval_bound_f32: bad arguments");
    const long long total = (long long)nparts * C;
    hipLaunchKernelGGL(bn_eval_bound_kernel, dim3(grid_for(total, 256 * 4, 1024)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(partials), total, C, scale, shift, relu, bound);
    return check_launch("trid_bn_eval_bound_f32");
}

extern "C" int trid_bn_eval_coeffs_f32(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, float* scale, float* shift, int C,
                                       void* stream) {
    TRID_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "trid_bn_eval_coeffs_f32: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, scale, shift, C);
    return check_launch("trid_bn_eval_coeffs_f32");
}

extern "C" int trid_bn_apply_f32(const float* y, const float* scale, const float* shift, const float* res,
                                 const float* rscale, const float* rshift, float* out, long long M, int C, int relu,
                                 uint64_t* relu_mask, float* amax, void* stream) {
    TRID_REQUIRE(y && scale && shift && out && M > 0 && C > 0 && C % 4 == 0, "trid_bn_apply_f32: bad arguments (C%%4)");
    TRID_REQUIRE((rscale == nullptr) == (rshift == nullptr), "trid_bn_apply_f32: rscale/rshift both or none");
    TRID_REQUIRE(aligned16(y) && aligned16(out) && aligned16(scale) && aligned16(shift) && (!res || aligned16(res)), "trid_bn_apply_f32: 16-byte alignment");
    const long long total4 = M * (C / 4);
    hipLaunchKernelGGL((bn_apply_kernel<0, 0>), dim3(grid_for(total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)y, (const float4*)scale, (const float4*)shift, (const float4*)res,
                       (const float4*)rscale, (const float4*)rshift, (float4*)out, total4, C / 4, relu,
                       (unsigned long long*)relu_mask, amax, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (const float*)nullptr, 0, stream_nt(total4 * 16));
    return check_launch("trid_bn_apply_f32");
}

extern "C" int trid_bn_apply_p16_f32(const void* y, int y_fmt, const float* scale, const float* shift, const void* res,
                                     const float* rscale, const float* rshift, int res_fmt, const float* res_amax, void* out,
                                     int fmt, long long M, int C, int relu, uint64_t* relu_mask, const float* bound_a,
                                     const float* bound_b, float* bound_sum, void* stream) {
    TRID_REQUIRE(y && scale && shift && out && (fmt == 2 || bound_a) && (fmt == 1 || fmt == 2) && M > 0 && C > 0 && C % 32 == 0,
                 "trid_bn_apply_p16_f32: bad arguments (fmt 1 / 2, C%%32)");
    TRID_REQUIRE(res_fmt >= 0 && res_fmt <= 2 && (res_fmt != 1 || res_amax), "trid_bn_apply_p16_f32: bad residual format");
    TRID_REQUIRE((rscale == nullptr) == (rshift == nullptr), "trid_bn_apply_p16_f32: rscale/rshift both or none");
    TRID_REQUIRE(!(res_fmt == 1 && rscale), "trid_bn_apply_p16_f32: a P16 residual is an identity residual (no BatchNorm on it)");
    TRID_REQUIRE(y_fmt == 0 || (y_fmt == 2 && fmt == 2), "trid_bn_apply_p16_f32: y_fmt is 0, or 2 in the bf16 mode (fmt 2)");
    TRID_REQUIRE(aligned16(y) && aligned16(out) && aligned16(scale) && aligned16(shift) && (!res || aligned16(res)), "trid_bn_apply_p16_f32: 16-byte alignment");
    const long long total4 = M * (C / 4);
    const dim3 grid(grid_for(total4, 256 * 4));
#define TRID_BN_APPLY_P16(OF, RF)                                                                                           \
    hipLaunchKernelGGL((bn_apply_kernel<OF, RF>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)y,              \
                       (const float4*)scale, (const float4*)shift, (const float4*)res, (const float4*)rscale,              \
                       (const float4*)rshift, (float4*)out, total4, C / 4, relu, (unsigned long long*)relu_mask,           \
                       (float*)nullptr, bound_a, bound_b, bound_sum, res_amax, y_fmt, stream_nt(total4 * 16))
    if (fmt == 1) {
        if (res_fmt == 1) TRID_BN_APPLY_P16(1, 1); else TRID_BN_APPLY_P16(1, 0);
    } else {
        if (res_fmt == 2) TRID_BN_APPLY_P16(2, 2); else TRID_BN_APPLY_P16(2, 0);
    }
#undef TRID_BN_APPLY_P16
    return check_launch("trid_bn_apply_p16_f32");
}

extern "C" int trid_bn_apply_pool2_f32(const float* y, const float* scale, const float* shift, float* out, int B, int H,
                                       int W, int C, int relu, float* amax, void* stream) {
    TRID_REQUIRE(y && out && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "trid_bn_apply_pool2_f32: bad arguments");
    TRID_REQUIRE((scale == nullptr) == (shift == nullptr), "trid_bn_apply_pool2_f32: scale/shift both or none");
    const long long total4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL((bn_apply_pool2_kernel<0, 0>), dim3(grid_for(total4, 256 * 2)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)y, (const float4*)scale, (const float4*)shift, (float4*)out, B, H, W, C / 4, relu,
                       total4, amax, (const float*)nullptr, (const float*)nullptr);
    return check_launch("trid_bn_apply_pool2_f32");
}

extern "C" int trid_bn_apply_pool2_p16_f32(const void* y, const float* scale, const float* shift, int in_fmt, const float* in_amax,
                                           void* out, int fmt, int B, int H, int W, int C, int relu, const float* bound,
                                           void* stream) {
    TRID_REQUIRE(y && out && (fmt == 2 || bound) && (fmt == 1 || fmt == 2) && (in_fmt == 0 || in_fmt == fmt) && (in_fmt != 1 || in_amax) &&
                     B > 0 && H % 2 == 0 && W % 2 == 0 && C % 32 == 0,
                 "trid_bn_apply_pool2_p16_f32: bad arguments (fmt 1 / 2, C%%32)");
    TRID_REQUIRE((scale == nullptr) == (shift == nullptr), "trid_bn_apply_pool2_p16_f32: scale/shift both or none");
    const long long total4 = (long long)B * (H / 2) * (W / 2) * (C / 4);
    const dim3 grid(grid_for(total4, 256 * 2));
#define TRID_BN_POOL_P16(OF, IF)                                                                                            \
    hipLaunchKernelGGL((bn_apply_pool2_kernel<OF, IF>), grid, dim3(256), 0, (hipStream_t)stream, (const float4*)y,        \
                       (const float4*)scale, (const float4*)shift, (float4*)out, B, H, W, C / 4, relu, total4,             \
                       (float*)nullptr, bound, in_amax)
    if (fmt == 1) {
        if (in_fmt == 1) TRID_BN_POOL_P16(1, 1); else TRID_BN_POOL_P16(1, 0);
    } else {
        if (in_fmt == 2) TRID_BN_POOL_P16(2, 2); else TRID_BN_POOL_P16(2, 0);
    }
#undef TRID_BN_POOL_P16
    return check_launch("trid_bn_apply_pool2_p16_f32");
}

extern "C" int trid_avgpool2_bwd_f32(const void* g, void* dx, int B, int H, int W, int C, int accumulate, int fmt,
                                     void* stream) {
    TRID_REQUIRE(g && dx && B > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && (fmt == 0 || fmt == 2), "trid_avgpool2_bwd_f32: bad arguments");
    const long long total4 = (long long)B * H * W * (C / 4);
    if (fmt == 2)
        hipLaunchKernelGGL(avgpool2_bwd_kernel<2>, dim3(grid_for(total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)g, (float4*)dx, B, H, W, C / 4, accumulate, total4);
    else
        hipLaunchKernelGGL(avgpool2_bwd_kernel<0>, dim3(grid_for(total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float4*)g, (float4*)dx, B, H, W, C / 4, accumulate, total4);
    return check_launch("trid_avgpool2_bwd_f32");
}

extern "C" long long trid_bn_bwd_ws_floats(int C) {
    const int CQ = C / 4;
    const int CW = CQ < 256 ? CQ : 256;
    return 2 * (long long)(1024 + 8) * CW * 8;  // sums + (trid_bn_bwd_reduce_bound_f32) maxima
}

// partials of the fused form (gemm_p16.hip, BnBwdFuse): one per 128-row tile and channel-quad slice
extern "C" long long trid_bn_bwd_fused_ws_floats(long long M, int C) {
    const int CQ = C / 4;
    const int CW = CQ < 256 ? CQ : 256;
    const long long S = CQ / CW;
    return ((M + 127) / 128 * S + 8) * CW * 8;
}

extern "C" int trid_bn_bwd_final_f32(const float* ws, const float* ws2, long long M, int C, const float* scale, float* dgamma,
                                     float* dbeta, float* bound, void* stream) {
    TRID_REQUIRE(ws && ws2 && scale && dgamma && dbeta && bound && M > 0 && C > 0 && C % 4 == 0, "trid_bn_bwd_final_f32: bad arguments");
    const int CQ = C / 4;
    TRID_REQUIRE(256 % CQ == 0 || CQ % 256 == 0, "trid_bn_bwd_final_f32: C/4 must divide 256 or be a multiple of 256 (C=%d)", C);
    const int CW = CQ < 256 ? CQ : 256;
    const int nblk = (int)((M + 127) / 128) * (CQ / CW);
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(CQ), dim3(256), 0, (hipStream_t)stream, ws, nblk, CQ, dgamma, dbeta, C, ws2,
                       scale, 1.f / (float)M, bound);
    return check_launch("trid_bn_bwd_final_f32");
}

static int bn_bwd_fill(BnBwdArgs& a, const float* g, const float* y, const float* act, const float* mean,
                       const float* invstd, const float* scale, const float* shift, int mask_mode, int pooled, int B,
                       int H, int W, int C, int g_fmt = 0, int y_fmt = 0) {
    TRID_REQUIRE(g && y && mean && invstd && scale && shift, "bn_bwd: null pointer");
    TRID_REQUIRE((g_fmt == 0 || g_fmt == 2) && (y_fmt == 0 || y_fmt == 2), "bn_bwd: g_fmt / y_fmt must be 0 (fp32) or 2 (bf16)");
    TRID_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "bn_bwd: bad shape");
    const int CQ = C / 4;
    TRID_REQUIRE(256 % CQ == 0 || CQ % 256 == 0, "bn_bwd: C/4 must divide 256 or be a multiple of 256 (C=%d)", C);
    TRID_REQUIRE(mask_mode >= 0 && mask_mode <= 3 && (mask_mode < 2 || act), "bn_bwd: bad mask mode");
    TRID_REQUIRE(!pooled || (H % 2 == 0 && W % 2 == 0), "bn_bwd: pooled needs even H,W");
    TRID_REQUIRE((long long)B * H * W < (1LL << 31), "bn_bwd: too many pixels");
    a.g = (const float4*)g; a.y = (const float4*)y; a.act = (const float4*)act;
    a.mean = (const float4*)mean; a.invstd = (const float4*)invstd;
    a.scale = (const float4*)scale; a.shift = (const float4*)shift;
    a.mask_mode = mask_mode; a.pooled = pooled;
    a.g_fmt = g_fmt;
    a.y_fmt = y_fmt;
    a.B = B; a.H = H; a.W = W; a.CQ = CQ;
    a.total4 = (long long)B * H * W * CQ;
    a.fdW = make_fastdiv((uint32_t)W);
    a.fdH = make_fastdiv((uint32_t)H);
    a.nt = stream_nt(a.total4 * 16);
    return TRID_OK;
}

extern "C" int trid_bn_bwd_reduce_f32(const float* g, const float* y, const float* act, const float* mean,
                                      const float* invstd, const float* scale, const float* shift, int mask_mode,
                                      int pooled, int B, int H, int W, int C, float* dgamma, float* dbeta, float* ws,
                                      void* stream) {
    return trid_bn_bwd_reduce_g_f32(g, 0, y, 0, act, mean, invstd, scale, shift, mask_mode, pooled, B, H, W, C, dgamma, dbeta, ws, stream);
}

extern "C" int trid_bn_bwd_reduce_g_f32(const void* g, int g_fmt, const void* y, int y_fmt, const float* act, const float* mean,
                                        const float* invstd, const float* scale, const float* shift, int mask_mode,
                                        int pooled, int B, int H, int W, int C, float* dgamma, float* dbeta, float* ws,
                                        void* stream) {
    BnBwdArgs a;
    int rc = bn_bwd_fill(a, (const float*)g, (const float*)y, act, mean, invstd, scale, shift, mask_mode, pooled, B, H, W, C, g_fmt, y_fmt);
    if (rc) return rc;
    TRID_REQUIRE(dgamma && dbeta && ws, "trid_bn_bwd_reduce_f32: null output");
    const int grid = bn_bwd_grid(a.total4, a.CQ);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, ws, (float*)nullptr);
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(a.CQ), dim3(256), 0, (hipStream_t)stream, ws, grid, a.CQ, dgamma,
                       dbeta, C, (const float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr);
    return check_launch("trid_bn_bwd_reduce_f32");
}

extern "C" int trid_bn_bwd_reduce_bound_f32(const float* g, const float* y, const float* act, const float* mean,
                                            const float* invstd, const float* scale, const float* shift, int mask_mode,
                                            int pooled, int B, int H, int W, int C, float* dgamma, float* dbeta, float* ws,
                                            float* bound, void* stream) {
    BnBwdArgs a;
    int rc = bn_bwd_fill(a, g, y, act, mean, invstd, scale, shift, mask_mode, pooled, B, H, W, C);
    if (rc) return rc;
    TRID_REQUIRE(dgamma && dbeta && ws && bound, "trid_bn_bwd_reduce_bound_f32: null output");
    const int grid = bn_bwd_grid(a.total4, a.CQ);
    const int CW = a.CQ < 256 ? a.CQ : 256;
    float* ws2 = ws + (long long)(1024 + 8) * CW * 8;
    const float invM = 1.f / (float)((long long)B * H * W);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, ws, ws2);
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(a.CQ), dim3(256), 0, (hipStream_t)stream, ws, grid, a.CQ, dgamma,
                       dbeta, C, (const float*)ws2, scale, invM, bound);
    return check_launch("trid_bn_bwd_reduce_bound_f32");
}

static int bn_bwd_dual_fill(BnBwdDual& a, const float* g, const uint64_t* bits, const float* y1, const float* y2, const float* mean1,
                            const float* invstd1, const float* scale1, const float* mean2, const float* invstd2, const float* scale2,
                            long long M, int C) {
    TRID_REQUIRE(g && bits && y1 && y2 && mean1 && invstd1 && scale1 && mean2 && invstd2 && scale2, "bn_bwd_dual: null pointer");
    TRID_REQUIRE(M > 0 && M < (1ll << 31) && C > 0 && C % 32 == 0, "bn_bwd_dual: bad shape (C %% 32)");
    const int CQ = C / 4;
    TRID_REQUIRE(256 % CQ == 0 || CQ % 256 == 0, "bn_bwd_dual: C/4 must divide 256 or be a multiple of 256 (C=%d)", C);
    a.g = (const float4*)g; a.bits = (const unsigned long long*)bits; a.y1 = (const float4*)y1; a.y2 = (const float4*)y2;
    a.mean1 = (const float4*)mean1; a.invstd1 = (const float4*)invstd1; a.scale1 = (const float4*)scale1;
    a.mean2 = (const float4*)mean2; a.invstd2 = (const float4*)invstd2; a.scale2 = (const float4*)scale2;
    a.CQ = CQ;
    a.total4 = M * CQ;
    a.nt = stream_nt(a.total4 * 16);
    return TRID_OK;
}

// ws: 2 x trid_bn_bwd_ws_floats(C) floats (one workspace per layer)
extern "C" int trid_bn_bwd_dual_reduce_bound_f32(const float* g, const uint64_t* relu_bits, const float* y1, const float* y2, const float* mean1,
                                                 const float* invstd1, const float* scale1, const float* mean2, const float* invstd2,
                                                 const float* scale2, long long M, int C, float* dgamma1, float* dgamma2, float* dbeta,
                                                 float* dbeta2, float* ws, float* bound1, float* bound2, void* stream) {
    BnBwdDual a;
    int rc = bn_bwd_dual_fill(a, g, relu_bits, y1, y2, mean1, invstd1, scale1, mean2, invstd2, scale2, M, C);
    if (rc) return rc;
    TRID_REQUIRE(dgamma1 && dgamma2 && dbeta && dbeta2 && ws && bound1 && bound2, "trid_bn_bwd_dual_reduce_bound_f32: null output");
    const int grid = bn_bwd_grid(a.total4, a.CQ);
    const int CW = a.CQ < 256 ? a.CQ : 256;
    const long long half = (long long)(1024 + 8) * CW * 8;
    float* wsA = ws; float* ws2A = ws + half; float* wsB = ws + 2 * half; float* ws2B = ws + 3 * half;
    const float invM = 1.f / (float)M;
    hipLaunchKernelGGL(bn_bwd_dual_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, wsA, ws2A, wsB, ws2B);
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(a.CQ), dim3(256), 0, (hipStream_t)stream, (const float*)wsA, grid, a.CQ, dgamma1, dbeta, C,
                       (const float*)ws2A, scale1, invM, bound1);
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(a.CQ), dim3(256), 0, (hipStream_t)stream, (const float*)wsB, grid, a.CQ, dgamma2, dbeta2, C,
                       (const float*)ws2B, scale2, invM, bound2);
    return check_launch("trid_bn_bwd_dual_reduce_bound_f32");
}

extern "C" int trid_bn_bwd_dual_apply_p16_f32(const float* g, const uint64_t* relu_bits, const float* y1, const float* y2, const float* mean1,
                                              const float* invstd1, const float* scale1, const float* mean2, const float* invstd2,
                                              const float* scale2, const float* dgamma1, const float* dgamma2, const float* dbeta, long long M,
                                              int C, void* dy1, void* dy2, const float* bound1, const float* bound2, void* stream) {
    BnBwdDual a;
    int rc = bn_bwd_dual_fill(a, g, relu_bits, y1, y2, mean1, invstd1, scale1, mean2, invstd2, scale2, M, C);
    if (rc) return rc;
    TRID_REQUIRE(dgamma1 && dgamma2 && dbeta && dy1 && dy2 && bound1 && bound2, "trid_bn_bwd_dual_apply_p16_f32: null pointer");
    hipLaunchKernelGGL(bn_bwd_dual_apply_kernel, dim3(grid_for(a.total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a, (const float4*)dgamma1,
                       (const float4*)dgamma2, (const float4*)dbeta, 1.f / (float)M, (float4*)dy1, (float4*)dy2, bound1, bound2);
    return check_launch("trid_bn_bwd_dual_apply_p16_f32");
}

extern "C" int trid_bn_bwd_apply_f32(const float* g, const float* y, const float* act, const float* mean,
                                     const float* invstd, const float* scale, const float* shift, const float* dgamma,
                                     const float* dbeta, int mask_mode, int pooled, int B, int H, int W, int C,
                                     float* dy, float* dres, float* amax, void* stream) {
    BnBwdArgs a;
    int rc = bn_bwd_fill(a, g, y, act, mean, invstd, scale, shift, mask_mode, pooled, B, H, W, C);
    if (rc) return rc;
    TRID_REQUIRE(dgamma && dbeta && dy, "trid_bn_bwd_apply_f32: null pointer");
    const float invM = 1.f / (float)((long long)B * H * W);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(grid_for(a.total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a,
                       (const float4*)dgamma, (const float4*)dbeta, invM, (float4*)dy, (float4*)dres, amax, (const float*)nullptr);
    return check_launch("trid_bn_bwd_apply_f32");
}

extern "C" int trid_bn_bwd_apply_p16_f32(const void* g, int g_fmt, const void* y, int y_fmt, const float* act, const float* mean,
                                         const float* invstd, const float* scale, const float* shift, const float* dgamma,
                                         const float* dbeta, int mask_mode, int pooled, int B, int H, int W, int C,
                                         void* dy, int fmt, void* dres, const float* bound, void* stream) {
    BnBwdArgs a;
    int rc = bn_bwd_fill(a, (const float*)g, (const float*)y, act, mean, invstd, scale, shift, mask_mode, pooled, B, H, W, C, g_fmt, y_fmt);
    if (rc) return rc;
    TRID_REQUIRE(dgamma && dbeta && dy && (fmt == 2 || bound) && (fmt == 1 || fmt == 2) && C % 32 == 0,
                 "trid_bn_bwd_apply_p16_f32: null pointer, fmt not 1 / 2 or C %% 32 != 0");
    const float invM = 1.f / (float)((long long)B * H * W);
    if (fmt == 1)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(grid_for(a.total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a,
                           (const float4*)dgamma, (const float4*)dbeta, invM, (float4*)dy, (float4*)dres, (float*)nullptr, bound);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<2>, dim3(grid_for(a.total4, 256 * 4)), dim3(256), 0, (hipStream_t)stream, a,
                           (const float4*)dgamma, (const float4*)dbeta, invM, (float4*)dy, (float4*)dres, (float*)nullptr, bound);
    return check_launch("trid_bn_bwd_apply_p16_f32");
}
