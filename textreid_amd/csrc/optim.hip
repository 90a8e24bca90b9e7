// MoCo state kernels: multi-tensor momentum (EMA) update of the key encoders
// (head.py:73-94), ring-buffer enqueue with a device-resident pointer
// (head.py:96-109), and a multi-tensor Adam step with per-tensor lr / weight
// decay (lib/solver/build.py:6-40 creates one param group per tensor).
// All HBM-bound: one launch streams every tensor through a chunk table instead
// of 178 (EMA) / 183 (Adam) x several tiny launches.

#include "common.h"

namespace trid {

__global__ __launch_bounds__(256) void ema_multi_kernel(const uint64_t* __restrict__ k_ptrs,
                                                        const uint64_t* __restrict__ q_ptrs,
                                                        const int64_t* __restrict__ sizes,
                                                        const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_off, int chunk_len, float m,
                                                        float om) {
    // This file is compiled with -ffp-contract=off (Makefile): mul, mul, add stay separately
    // rounded, so the EMA is bit-identical to the reference's three torch ops.
    const int c = blockIdx.x;
    const int ti = chunk_tensor[c];
    const long long off = chunk_off[c];
    float* __restrict__ k = reinterpret_cast<float*>(k_ptrs[ti]) + off;
    const float* __restrict__ q = reinterpret_cast<const float*>(q_ptrs[ti]) + off;
    long long n = sizes[ti] - off;
    if (n > chunk_len) n = chunk_len;
    if (((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(q)) & 15u) == 0) {
        const long long n4 = n >> 2;
        for (long long i = threadIdx.x; i < n4; i += 256) {
            float4 a = reinterpret_cast<float4*>(k)[i];
            const float4 b = reinterpret_cast<const float4*>(q)[i];
            // param_k*m + param_q*(1-m): two rounded products, one rounded sum (no FMA
            // contraction), so the result is bit-identical to the reference's torch ops
            a.x = a.x * m + b.x * om; a.y = a.y * m + b.y * om;
            a.z = a.z * m + b.z * om; a.w = a.w * m + b.w * om;
            reinterpret_cast<float4*>(k)[i] = a;
        }
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) k[i] = k[i] * m + q[i] * om;
    } else {
        for (long long i = threadIdx.x; i < n; i += 256) k[i] = k[i] * m + q[i] * om;
    }
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const uint64_t* __restrict__ p_ptrs,
                                                         const uint64_t* __restrict__ g_ptrs,
                                                         const uint64_t* __restrict__ m_ptrs,
                                                         const uint64_t* __restrict__ v_ptrs,
                                                         const int64_t* __restrict__ sizes, const float* __restrict__ lrs,
                                                         const float* __restrict__ wds,
                                                         const int32_t* __restrict__ chunk_tensor,
                                                         const int64_t* __restrict__ chunk_off, int chunk_len, float b1,
                                                         float b2, float eps, const float* __restrict__ bc1s,
                                                         const float* __restrict__ bc2s, int decoupled) {
    const int c = blockIdx.x;
    const int ti = chunk_tensor[c];
    const long long off = chunk_off[c];
    float* __restrict__ p = reinterpret_cast<float*>(p_ptrs[ti]) + off;
    const float* __restrict__ g = reinterpret_cast<const float*>(g_ptrs[ti]) + off;
    float* __restrict__ mo = reinterpret_cast<float*>(m_ptrs[ti]) + off;
    float* __restrict__ vo = reinterpret_cast<float*>(v_ptrs[ti]) + off;
    long long n = sizes[ti] - off;
    if (n > chunk_len) n = chunk_len;
    const float lr = lrs[ti], wd = wds[ti];
    const float bc1 = bc1s[ti], bc2 = bc2s[ti];  // per tensor: torch.optim.Adam keeps one step count per parameter
    const float step = lr / bc1;       // lr / (1 - b1^t)
    const float inv_bc2s = 1.f / bc2;  // 1 / sqrt(1 - b2^t)
    for (long long i = threadIdx.x; i < n; i += 256) {
        float w = p[i];
        float gr = g[i];
        if (decoupled) w -= lr * wd * w; else gr = fmaf(wd, w, gr);
        const float m1 = b1 * mo[i] + (1.f - b1) * gr;
        const float v1 = b2 * vo[i] + (1.f - b2) * gr * gr;
        mo[i] = m1;
        vo[i] = v1;
        p[i] = w - step * m1 / (sqrtf(v1) * inv_bc2s + eps);
    }
}

__global__ void enqueue_kernel(float* __restrict__ vq, float* __restrict__ tq, int64_t* __restrict__ idq,
                               const int64_t* __restrict__ ptr, const float* __restrict__ vk,
                               const float* __restrict__ tk, const int64_t* __restrict__ ids, int K, int C, int B) {
    // The pointer comes from a buffer a checkpoint may have filled (another batch size, another world size): rows
    // are addressed modulo K, so a pointer that is not a multiple of B wraps around instead of writing past the end.
    const long long p = ((ptr[0] % K) + K) % K;
    const long long total = (long long)B * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / C, c = i - row * C;
        long long dst = p + row;
        if (dst >= K) dst -= K;
        vq[dst * C + c] = vk[i];
        tq[dst * C + c] = tk[i];
        if (c == 0) idq[dst] = ids[row];
    }
}
__global__ void enqueue_advance_kernel(int64_t* ptr, int K, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ptr[0] = ((ptr[0] % K) + K + B) % K;
}

}  // namespace trid

using namespace trid;

extern "C" int trid_ema_multi_f32(const uint64_t* k_ptrs, const uint64_t* q_ptrs, const int64_t* sizes,
                                  const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_len,
                                  float m, float one_minus_m, void* stream) {
    TRID_REQUIRE(k_ptrs && q_ptrs && sizes && chunk_tensor && chunk_off && n_chunks > 0 && chunk_len > 0 && chunk_len % 4 == 0,
                 "trid_ema_multi_f32: bad arguments");
    hipLaunchKernelGGL(ema_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, k_ptrs, q_ptrs, sizes,
                       chunk_tensor, chunk_off, chunk_len, m, one_minus_m);
    return check_launch("trid_ema_multi_f32");
}

extern "C" int trid_adam_multi_f32(const uint64_t* p_ptrs, const uint64_t* g_ptrs, const uint64_t* m_ptrs,
                                   const uint64_t* v_ptrs, const int64_t* sizes, const float* lrs, const float* wds,
                                   const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_len,
                                   float beta1, float beta2, float eps, const float* bias_c1, const float* bias_c2,
                                   int decoupled, void* stream) {
    TRID_REQUIRE(p_ptrs && g_ptrs && m_ptrs && v_ptrs && sizes && lrs && wds && chunk_tensor && chunk_off && bias_c1 && bias_c2,
                 "trid_adam_multi_f32: null pointer");
    TRID_REQUIRE(n_chunks > 0 && chunk_len > 0, "trid_adam_multi_f32: bad arguments");
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, p_ptrs, g_ptrs, m_ptrs, v_ptrs,
                       sizes, lrs, wds, chunk_tensor, chunk_off, chunk_len, beta1, beta2, eps, bias_c1, bias_c2, decoupled);
    return check_launch("trid_adam_multi_f32");
}

extern "C" int trid_enqueue_f32(float* v_queue, float* t_queue, int64_t* id_queue, int64_t* ptr, const float* v_keys,
                                const float* t_keys, const int64_t* ids, int K, int C, int B, void* stream) {
    TRID_REQUIRE(v_queue && t_queue && id_queue && ptr && v_keys && t_keys && ids, "trid_enqueue_f32: null pointer");
    TRID_REQUIRE(K > 0 && C > 0 && B > 0 && K % B == 0, "trid_enqueue_f32: K (%d) must be a multiple of the batch (%d)", K, B);  // head.py:101
    hipLaunchKernelGGL(enqueue_kernel, dim3(grid_for((long long)B * C, 256, 256)), dim3(256), 0, (hipStream_t)stream,
                       v_queue, t_queue, id_queue, ptr, v_keys, t_keys, ids, K, C, B);
    hipLaunchKernelGGL(enqueue_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ptr, K, B);
    return check_launch("trid_enqueue_f32");
}
