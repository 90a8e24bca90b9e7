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

}  // namespace trid

using namespace trid;

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
