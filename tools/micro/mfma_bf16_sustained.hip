// Sustained (power-limited) bf16 MFMA rate: v_mfma_f32_32x32x16_bf16 from registers for ~1 s per case,
// with (a) constant operands and (b) pseudo-random bf16 operands re-read from LDS every 48 MFMAs.
// The burst figure of mfma_bf16_peak.hip (2480 TFLOP/s over 10 ms) is a clock-boosted number; this
// probe shows what the box holds once the power controller has settled.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ uint4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned s = 1234567u + tid * 2654435761u + blockIdx.x * 97u;
    for (int i = tid; i < 4096; i += 512) {
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u;
            // two bf16 values in [-2, 2) with random mantissas
            const unsigned a = 0x3f800000u | (s & 0x807f0000u), b = 0x3f800000u | ((s << 9) & 0x807f0000u);
            w[j] = RANDOM ? ((a >> 16) | (b & 0xffff0000u)) : 0x3f803f80u;
        }
        lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        uint4 ua[3], ub[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) { ua[g] = lds[(lane + 64 * g + 7 * i) & 4095]; ub[g] = lds[(lane + 64 * g + 2048 + 11 * i) & 4095]; }
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, ua[g % 3]), b = __builtin_bit_cast(bf16x8, ub[(g / 3) % 3]);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        // keep magnitudes bounded without leaving the MFMA-dominated regime
        if ((i & 255) == 255) { c0 *= 1e-3f; c1 *= 1e-3f; c2 *= 1e-3f; c3 *= 1e-3f; }
    }
    float r = 0;
    for (int j = 0; j < 16; ++j) r += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * 512 + tid] = r;
}

template <int RANDOM>
static void run(const char* name) {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    const int blocks = 512, iters = 400000;  // 2 WG/CU, ~1 s
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<RANDOM>, dim3(blocks), dim3(512), 0, 0, out, 1000);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<RANDOM>, dim3(blocks), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 8 * iters * 48 * 2.0 * 32 * 32 * 16;
        printf("%-28s %8.1f ms  %7.1f TFLOP/s bf16 sustained\n", name, ms, flops / ms / 1e9);
    }
    (void)hipFree(out);
}
int main() {
    run<0>("constant operands");
    run<1>("random bf16 operands");
    return 0;
}
