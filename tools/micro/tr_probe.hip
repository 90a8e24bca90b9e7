// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds u16 value = element index; lane l reads 8 bytes
// at byte address 8*l (a wave covers 512 B); prints the four 16-bit values each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out, int stride) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane -> (row = (l&15)/4 within a 4-row tile, col chunk = l&3), tile = l>>4; rows `stride` bytes apart
    unsigned addr = (unsigned)(uintptr_t)lds + ((l & 15) >> 2) * stride + (l & 3) * 8 + (l >> 4) * 32;
    v2u r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[2 * l] = r.x;
    out[2 * l + 1] = r.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8);
    for (int stride : {32, 128, 512}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d bytes (elements: row r, col c -> r*%d + c)\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u\n", l, h[2*l] & 0xffff, h[2*l] >> 16, h[2*l+1] & 0xffff, h[2*l+1] >> 16);
    }
    return 0;
}
