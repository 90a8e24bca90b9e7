// Shared host/device helpers for libtextreid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/textreid_hip.h"

#define TRID_WAVE 64

namespace trid {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return TRID_OK;
}

#define TRID_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            trid::set_error(__VA_ARGS__);  \
            return TRID_E_INVALID;         \
        }                                  \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Division of a 32-bit unsigned by a runtime-constant divisor via multiply-high.
struct FastDiv {
    uint32_t mul, shift, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) {
        f.mul = 0;
        f.shift = 0;
        return f;
    }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;  // ceil(log2 d)
    uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    f.mul = (uint32_t)m;
    f.shift = l;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    if (f.d == 1) return n;
    uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, block b runs on
// XCD b % 8): each XCD gets a contiguous chunk of the logical id space, so
// neighbouring tiles (which share operand panels) hit the same private L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t nx = 8;
    if (nwg < 2 * nx) return bid;
    uint32_t q = nwg / nx, r = nwg % nx;
    uint32_t xcd = bid % nx, idx = bid / nx;
    uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

inline int grid_for(int64_t work_items, int per_block, int cap = 4096) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace trid
