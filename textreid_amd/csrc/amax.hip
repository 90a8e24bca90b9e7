// Largest magnitude of a tensor / of each tensor of a list, as DEVICE scalars: the per-tensor power-of-two
// operand scales of the fp16-split GEMM arithmetic (trid_gemm_desc.precision == 16, split_common.h).
// max|x| is order-independent, so the atomic fold is deterministic.  Non-negative IEEE floats order like
// their bit patterns: the fold is an unsigned integer atomicMax on the bits; `out` must start at 0.

#include "common.h"

namespace trid {

__device__ __forceinline__ unsigned absbits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }

__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    unsigned m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x4[i];
        const unsigned a = absbits(v.x), b = absbits(v.y), c = absbits(v.z), d = absbits(v.w);
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        const unsigned q = ab > cd ? ab : cd;
        m = q > m ? q : m;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const unsigned a = absbits(x[(n4 << 2) + threadIdx.x]);
        m = a > m ? a : m;
    }
    m = wave_max_u(m);
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
        atomicMax(out, a > b ? a : b);
    }
}

// gridDim.y workgroups per tensor, each over a contiguous slice; `out` must start at 0
__global__ __launch_bounds__(256) void amax_multi_kernel(const float* const* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                         unsigned* __restrict__ out) {
    const float* x = ptrs[blockIdx.x];
    const long long n = sizes[blockIdx.x];
    long long per = (n + gridDim.y - 1) / gridDim.y;
    per = (per + 3) & ~3LL;  // slices start on a 16-byte boundary when the tensor does
    const long long lo = (long long)blockIdx.y * per;
    const long long hi = lo + per < n ? lo + per : n;
    if (lo >= hi) return;
    unsigned m = 0;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long q_lo = lo >> 2, q_hi = hi >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long long i = q_lo + threadIdx.x; i < q_hi; i += 256) {
            const float4 v = x4[i];
            const unsigned a = absbits(v.x), b = absbits(v.y), c = absbits(v.z), d = absbits(v.w);
            const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
            const unsigned q = ab > cd ? ab : cd;
            m = q > m ? q : m;
        }
        for (long long i = (q_hi << 2) + threadIdx.x; i < hi; i += 256) {
            const unsigned a = absbits(x[i]);
            m = a > m ? a : m;
        }
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            const unsigned a = absbits(x[i]);
            m = a > m ? a : m;
        }
    }
    m = wave_max_u(m);
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
        const unsigned r = a > b ? a : b;
        if (r != 0) atomicMax(out + blockIdx.x, r);
    }
}

// Eval-mode output bounds (gemm_common.h EvalBound): per convolution i the pair
//     coef[2i] = max_n |scale_n| * ||w_n||_1,   coef[2i + 1] = max_n |shift_n|
// so that |scale_n (w_n . x) + shift_n| <= coef[2i] * max|x| + coef[2i + 1] for every output channel n.  table[i] =
// {w, N, K, scale, shift} (device, 5 x int64): w fp32 [N][K] with each output channel's K weights contiguous (both the
// OIHW and the OHWI layouts), scale / shift [N] the eval-mode BatchNorm coefficients.  One wave per output channel;
// `coef` must start at 0 (integer atomicMax on non-negative floats).
__global__ __launch_bounds__(256) void eval_bound_coefs_kernel(const long long* __restrict__ table, unsigned* __restrict__ coef) {
    const long long* e = table + 5 * (long long)blockIdx.y;
    const float* __restrict__ w = reinterpret_cast<const float*>(e[0]);
    const int N = (int)e[1], K = (int)e[2];
    const float* __restrict__ scale = reinterpret_cast<const float*>(e[3]);
    const float* __restrict__ shift = reinterpret_cast<const float*>(e[4]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned c1 = 0, c2 = 0;
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += fabsf(w[(long long)n * K + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const unsigned a = absbits(scale[n] * s * 1.0001f), b = absbits(shift[n]);  // (the slack covers the order of the sum)
        c1 = a > c1 ? a : c1;
        c2 = b > c2 ? b : c2;
    }
    if (lane == 0) {
        if (c1 != 0) atomicMax(coef + 2 * blockIdx.y, c1);
        if (c2 != 0) atomicMax(coef + 2 * blockIdx.y + 1, c2);
    }
}

}  // namespace trid

using namespace trid;

extern "C" int trid_eval_bound_coefs_f32(const long long* table, int n_tensors, float* coef, void* stream) {
    TRID_REQUIRE(table && coef && n_tensors > 0, "trid_eval_bound_coefs_f32: bad arguments");
    hipLaunchKernelGGL(eval_bound_coefs_kernel, dim3(32, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, table,
                       reinterpret_cast<unsigned*>(coef));
    return check_launch("trid_eval_bound_coefs_f32");
}

extern "C" int trid_amax_f32(const float* x, long long n, float* out, void* stream) {
    TRID_REQUIRE(x && out && n > 0 && aligned16(x), "trid_amax_f32: bad arguments (x must be 16-byte aligned)");
    hipLaunchKernelGGL(amax_kernel, dim3(grid_for(n / 4 + 1, 256 * 8, 1024)), dim3(256), 0, (hipStream_t)stream, x, n,
                       reinterpret_cast<unsigned*>(out));
    return check_launch("trid_amax_f32");
}

extern "C" int trid_amax_multi_f32(const float* const* ptrs, const long long* sizes, int n_tensors, float* out, void* stream) {
    TRID_REQUIRE(ptrs && sizes && out && n_tensors > 0, "trid_amax_multi_f32: bad arguments");
    hipLaunchKernelGGL(amax_multi_kernel, dim3(n_tensors, 16), dim3(256), 0, (hipStream_t)stream, ptrs, sizes,
                       reinterpret_cast<unsigned*>(out));
    return check_launch("trid_amax_multi_f32");
}
