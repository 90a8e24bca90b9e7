// bf16 MFMA probes: what does v_mfma_f32_32x32x16_bf16 sustain on this box
//   (a) from registers only, (b) fed by ds_read_b128 fragment reads at the split-GEMM ratio
//   (24 reads per 48 MFMAs), (c) with a co-resident wave per SIMD doing plain VALU work?
// usage: ./mfma_bf16_peak   (prints TFLOP/s of bf16 MFMA work and the implied clock-normalised utilisation)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>  // 0: registers only, 1: + LDS fragment reads, 2: half the waves do VALU instead
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ uint4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = make_uint4(0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    uint4 ua = lds[lane], ub = lds[64 + lane];
    float s = 0;
    if (MODE == 2 && wave >= 4) {  // VALU partner: ~190 dependent-free VALU ops per "K-tile"
        float x0 = tid, x1 = tid + 1, x2 = tid + 2, x3 = tid + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 48; ++j) {
                x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0002f + 0.25f; x2 = x2 * 0.9999f + 0.125f; x3 = x3 * 0.9998f + 1.f;
            }
        }
        s = x0 + x1 + x2 + x3;
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                if (MODE == 1) {  // 2 fragment reads per 4 MFMAs
                    ua = lds[(lane + 64 * g + i) & 4095];
                    ub = lds[(lane + 64 * g + 2048 + i) & 4095];
                }
                const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, int mfma_waves) {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 50);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * mfma_waves * iters * 48 * 2.0 * 32 * 32 * 16;
        printf("%-58s %d WG/CU: %7.2f ms  %7.1f TFLOP/s bf16\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<0>("registers only, 8 MFMA waves per WG", 1, 8);
    run<0>("registers only, 8 MFMA waves per WG", 2, 8);
    run<1>("+24 ds_read_b128 per 48 MFMA, 8 MFMA waves per WG", 1, 8);
    run<1>("+24 ds_read_b128 per 48 MFMA, 8 MFMA waves per WG", 2, 8);
    run<2>("4 MFMA waves + 4 VALU waves (192 v_fma per 48 MFMA)", 1, 4);
    run<2>("4 MFMA waves + 4 VALU waves (192 v_fma per 48 MFMA)", 2, 4);
    return 0;
}
