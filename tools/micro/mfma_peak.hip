// Pure-MFMA fp32 peak probe: what does v_mfma_f32_32x32x2_f32 sustain on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int wpb = 1; wpb <= 2; ++wpb) {
        int blocks = 256 * wpb * 2, iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 2.f);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2.0 * 32 * 32 * 2;
            printf("blocks %d (%d per CU): %.2f ms  %.1f TFLOP/s\n", blocks, blocks / 256, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
