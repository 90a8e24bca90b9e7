// Helpers of the split-precision GEMM kernel (gemm_bf16.hip): fp32 -> bf16
// plane splitting while staging into LDS, LDS slot maps, branch-free predicated loads.
#pragma once
#include "gemm_common.h"

namespace trid {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS image of one operand tile with R rows: 16-byte slots (8 consecutive k of one row), one image per plane.
//   K-contiguous operands:  slot = kgroup*(R+1) + row                      (loader lane = one slot)
//   M/N-contiguous operands (R = 128): slot = kgroup*144 + (row%4)*36 + row/4
//     -> a lane that loaded a float4 ALONG the rows writes its 4 rows to 4 slot runs that are
//        consecutive across lanes, and MFMA fragment reads (32 consecutive rows per half-wave) stay
//        conflict-free for ds_read_b128's 16-lane groups ((row%4)*4 + row/4 is distinct mod 16).
#ifndef TRID_KC_PAD
#define TRID_KC_PAD 2  // row pitch of a K-contiguous plane = R + 2 slots: see slot_of
#endif
__host__ __device__ constexpr int plane_slots(int R) { return 4 * (R + TRID_KC_PAD) > 4 * 144 ? 4 * (R + TRID_KC_PAD) : 4 * 144; }
template <bool KCONTIG, int R>
__device__ __forceinline__ int slot_of(int kg, int row) {
    // pitch R + 2: the loader's eight-lane store groups hold (kg 0..3) x (2 rows); with a pitch = 2 (mod 8) slots their
    // eight 16-byte slots fall into eight different 4-bank groups of the 32-bank store path (pitch R + 1 made
    // (kg+1, row) collide with (kg, row+1): SQ_LDS_BANK_CONFLICT was 28 % of the LDS-active cycles)
    if (KCONTIG) return kg * (R + TRID_KC_PAD) + row;
    return kg * 144 + (row & 3) * 36 + (row >> 2);
}

// two fp32 -> one dword of two bf16 (round-to-nearest-even); `lo` lands in bits 0..15
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NPL, int PLANE>
__device__ __forceinline__ void split_store(const float (&v)[8], uint4* __restrict__ dst) {
    // dst: plane 0 slot; planes are PLANE slots apart.  Per pair of values: one packed convert
    // per plane, residuals formed exactly in fp32 (x - float(bf16(x)) is representable).
    unsigned w[NPL][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2 r = {v[2 * q], v[2 * q + 1]};
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const unsigned h = cvt_pk_bf16(r.x, r.y);
            w[pl][q] = h;
            if (pl + 1 < NPL) {
                r.x -= __builtin_bit_cast(float, h << 16);  // (a packed v_pk_add_f32 here measured 2 % slower)
                r.y -= __builtin_bit_cast(float, h & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) dst[pl * PLANE] = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

// Wide loader store: this lane holds w[kk] = float4 along 4 consecutive rows for k = 4kq+kk.  Each row's
// 4 k-values become half a slot (8 bytes) per plane in the swizzled M/N-contiguous image.
template <int NPL, int PLANE>
__device__ __forceinline__ void split_store_wide(const float4 (&w)[4], uint4* __restrict__ base, int mq, int kq) {
    const int kg = kq >> 1, half = kq & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r0, r1, r2, r3;
        if (j == 0) { r0 = w[0].x; r1 = w[1].x; r2 = w[2].x; r3 = w[3].x; }
        else if (j == 1) { r0 = w[0].y; r1 = w[1].y; r2 = w[2].y; r3 = w[3].y; }
        else if (j == 2) { r0 = w[0].z; r1 = w[1].z; r2 = w[2].z; r3 = w[3].z; }
        else { r0 = w[0].w; r1 = w[1].w; r2 = w[2].w; r3 = w[3].w; }
        uint2* dst = reinterpret_cast<uint2*>(base + slot_of<false, 128>(kg, 4 * mq + j)) + half;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const unsigned h0 = cvt_pk_bf16(r0, r1), h1 = cvt_pk_bf16(r2, r3);
            dst[pl * PLANE * 2] = make_uint2(h0, h1);
            if (pl + 1 < NPL) {
                r0 -= __builtin_bit_cast(float, h0 << 16);
                r1 -= __builtin_bit_cast(float, h0 & 0xffff0000u);
                r2 -= __builtin_bit_cast(float, h1 << 16);
                r3 -= __builtin_bit_cast(float, h1 & 0xffff0000u);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fp16 two-plane split ("f16x3": 3 MFMA products per multiply-add).  x' = x * scale (per-tensor power of
// two that puts max|x'| in [2^13, 2^14), inside fp16's range with 4x headroom), hi = fp16(x'),
// lo = fp16(x' - hi) (the residual is exact in fp32; |lo| <= 2^-11 |x'|):
// |x' - hi - lo| <= 2^-22 |x'| wherever the residual is a NORMAL fp16 number, i.e. for |x'| >= 2^-3 = 2^-16 of
// the tensor maximum; smaller elements keep an ABSOLUTE error <= 2^-25 (fp16's subnormal quantum; 2^-14 if the
// matrix pipe flushed subnormal inputs), i.e. <= 2^-38 (2^-27) of the maximum - far below the 2^-22 of the
// maximum that the largest elements are allowed.  A product is hi*hi + hi*lo + lo*hi, all three in ONE fp32
// MFMA accumulator (same scale: no 2^11 factor, no second accumulator, one multiply less per element than a
// scaled residual); the dropped lo*lo term is <= 2^-22 |ab|.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// power-of-two scale for a tensor whose largest magnitude is `amax`: max|x * scale| in [2^13, 2^14)
__device__ __forceinline__ float f16_scale_of(float amax) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((bits >> 23) & 0xffu);           // biased exponent of amax (floor(log2) + 127)
    if (e == 0 || e == 0xff) return 1.f;                 // zero / denormal / inf / nan: leave unscaled
    return __builtin_bit_cast(float, (unsigned)(127 + 13 + 127 - e) << 23);  // 2^(13 - floor(log2 amax))
}

// two values -> (hi dword, lo dword) of packed fp16 pairs
__device__ __forceinline__ void f16_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    hi = cvt_pk_f16(x0, x1);
    const f16x2 h = __builtin_bit_cast(f16x2, hi);
    lo = cvt_pk_f16(x0 - (float)h.x, x1 - (float)h.y);
}

template <int PLANE>
__device__ __forceinline__ void split_store_f16(const float (&v)[8], float scale, uint4* __restrict__ dst) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f16_split2(v[2 * q] * scale, v[2 * q + 1] * scale, h[q], l[q]);
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[PLANE] = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int PLANE>
__device__ __forceinline__ void split_store_wide_f16(const float4 (&w)[4], float scale, uint4* __restrict__ base, int mq, int kq) {
    const int kg = kq >> 1, half = kq & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r0, r1, r2, r3;
        if (j == 0) { r0 = w[0].x; r1 = w[1].x; r2 = w[2].x; r3 = w[3].x; }
        else if (j == 1) { r0 = w[0].y; r1 = w[1].y; r2 = w[2].y; r3 = w[3].y; }
        else if (j == 2) { r0 = w[0].z; r1 = w[1].z; r2 = w[2].z; r3 = w[3].z; }
        else { r0 = w[0].w; r1 = w[1].w; r2 = w[2].w; r3 = w[3].w; }
        uint2* dst = reinterpret_cast<uint2*>(base + slot_of<false, 128>(kg, 4 * mq + j)) + half;
        unsigned h0, l0, h1, l1;
        f16_split2(r0 * scale, r1 * scale, h0, l0);
        f16_split2(r2 * scale, r3 * scale, h1, l1);
        dst[0] = make_uint2(h0, h1);
        dst[PLANE * 2] = make_uint2(l0, l1);
    }
}

// bound of an eval-mode epilogue's output (gemm_common.h EvalBound); the factor covers the fp32 rounding of the sums
__device__ __forceinline__ float eval_out_bound(const EvalBound& e) {
    float b = fmaf(e.coef[0] * 1.001f, *e.tin, e.coef[1]);
    if (e.tres != nullptr) b += *e.tres;
    return b;
}
// running maximum of |v| as the bit pattern of a non-negative float
__device__ __forceinline__ unsigned absmax4u(unsigned m, float4 v) {
    const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
    const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    const unsigned q = ab > cd ? ab : cd;
    return q > m ? q : m;
}
__device__ __forceinline__ unsigned wave_umax(unsigned m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(m, o, 64);
        m = t > m ? t : m;
    }
    return m;
}

// BatchNorm-backward partials (bn_pool.hip reduce kernels, gemm_p16.hip BnBwdFuse -> bn_bwd_reduce_final_kernel): float offset of
// the 8-float entry of partial block b (of nblk; block b covers the channel-quad slice b % S, S = CQ / CW) for quad ql of that
// slice.  The entries of ONE quad lie next to each other (quad-major): the fold, one workgroup per quad with a thread per
// partial, reads whole lines - in the block-major order of the first version every thread touched its own line, and the fold
// (55 launches per step on the backward's critical chain) took 12-28 us.
__device__ __host__ __forceinline__ long long bn_bwd_partial_index(int b, int ql, int nblk, int CQ) {
    const int CW = CQ < 256 ? CQ : 256, S = CQ / CW;
    const long long q = (long long)(b % S) * CW + ql;  // global quad
    return (q * (nblk / S) + b / S) * 8;
}

// ---------------------------------------------------------------- P16 (pre-split GEMM operand) element access
// A [rows][C] tensor in P16 (gemm_p16.hip): per row and 32-channel group 128 bytes = [hi x 32 | lo x 32] fp16 of
// x * 2^s.  These kernels work on channel QUADS (float4): quad cq of a row lives at 8-byte unit
// row * (C/2) + (cq / 8) * 16 + (cq % 8) (high parts) and 8 units further (low parts): a wave writes whole
// 64-byte plane halves, fully coalesced.
// Streaming accesses of the HBM-bound passes: `nt` (uniform per launch) selects non-temporal loads / stores.  Measured on
// the BatchNorm passes (tools/exp/nt_ab.sh, profiles/r04i_bn_nontemporal.txt): tensors of >= 200 MB - read once and written
// once per pass, far beyond what the 256 MB Infinity Cache can hand to the next kernel - gain 8-14 % (5.2-5.4 -> 5.9-6.6
// TB/s); tensors of <= 100 MB LOSE 5-12 % (their consumer finds part of them on chip).  The launchers set it by size.
typedef float nt_v4f __attribute__((ext_vector_type(4)));
constexpr long long STREAM_NT_MIN_BYTES = 128ll << 20;
__device__ __forceinline__ float4 ld_stream4(const float4* p, bool nt) {
    if (nt) {
        const nt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const nt_v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
__device__ __forceinline__ void st_stream4(float4* p, float4 v, bool nt) {
    if (nt) {
        const nt_v4f r = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(r, reinterpret_cast<nt_v4f*>(p));
    } else {
        *p = v;
    }
}
__device__ __forceinline__ void st_stream2(uint2* p, uint2 v, bool nt) {
    if (nt) __builtin_nontemporal_store(((unsigned long long)v.y << 32) | v.x, reinterpret_cast<unsigned long long*>(p));
    else *p = v;
}

__device__ __forceinline__ void p16_store4(uint2* __restrict__ base, long long i, int CQ, float4 v, float scale, bool nt = false) {
    const long long row = i / CQ;
    const int cq = (int)(i - row * CQ);
    unsigned h0, l0, h1, l1;
    f16_split2(v.x * scale, v.y * scale, h0, l0);
    f16_split2(v.z * scale, v.w * scale, h1, l1);
    uint2* dst = base + row * (2 * CQ) + (cq >> 3) * 16 + (cq & 7);
    st_stream2(dst, make_uint2(h0, h1), nt);
    st_stream2(dst + 8, make_uint2(l0, l1), nt);
}
// The same store from a wave whose lanes 2k, 2k + 1 hold quads i, i + 1 (i even) of one row - the elementwise passes, whose
// quad index runs with the thread index: the pair swaps plane halves (the even lane takes both quads' high parts, the odd lane
// both low parts) and every lane writes ONE 16-byte piece, so a wave's store instruction covers whole 128-byte lines.  With
// two 8-byte stores per lane each instruction wrote every other 64-byte half line: bn_apply ran at 4.6-4.75 TB/s writing P16
// against 5.5-6.9 writing fp32 (profiles/r05g_bn_bandwidth_probe.txt).  Every lane of the wave must make the call (shuffle).
// (row, cq) form: for callers that carry the element's row and channel quad along instead of dividing i by CQ per element
__device__ __forceinline__ void p16_store4_pair_rc(uint2* __restrict__ base, long long row, int cq, int CQ, float4 v, float scale, bool nt = false) {
    unsigned h0, l0, h1, l1;
    f16_split2(v.x * scale, v.y * scale, h0, l0);
    f16_split2(v.z * scale, v.w * scale, h1, l1);
    const bool odd = (cq & 1) != 0;
    const unsigned rx = __shfl_xor(odd ? h0 : l0, 1, 64), ry = __shfl_xor(odd ? h1 : l1, 1, 64);
    const float4 piece = odd ? make_float4(__uint_as_float(rx), __uint_as_float(ry), __uint_as_float(l0), __uint_as_float(l1))
                             : make_float4(__uint_as_float(h0), __uint_as_float(h1), __uint_as_float(rx), __uint_as_float(ry));
    uint2* dst = base + row * (2 * CQ) + (cq >> 3) * 16 + (odd ? 8 + (cq & 7) - 1 : (cq & 7));
    st_stream4(reinterpret_cast<float4*>(dst), piece, nt);
}
__device__ __forceinline__ void p16_store4_pair(uint2* __restrict__ base, long long i, int CQ, float4 v, float scale, bool nt = false) {
    const long long row = i / CQ;
    const int cq = (int)(i - row * CQ);
    unsigned h0, l0, h1, l1;
    f16_split2(v.x * scale, v.y * scale, h0, l0);
    f16_split2(v.z * scale, v.w * scale, h1, l1);
    const bool odd = (cq & 1) != 0;
    const unsigned rx = __shfl_xor(odd ? h0 : l0, 1, 64), ry = __shfl_xor(odd ? h1 : l1, 1, 64);
    const float4 piece = odd ? make_float4(__uint_as_float(rx), __uint_as_float(ry), __uint_as_float(l0), __uint_as_float(l1))
                             : make_float4(__uint_as_float(h0), __uint_as_float(h1), __uint_as_float(rx), __uint_as_float(ry));
    uint2* dst = base + row * (2 * CQ) + (cq >> 3) * 16 + (odd ? 8 + (cq & 7) - 1 : (cq & 7));
    st_stream4(reinterpret_cast<float4*>(dst), piece, nt);
}
// ... for a lane that already holds the planes of its channel quad (dwords q0 = channels 0, 1; q1 = channels 2, 3) and the
// byte address `hi` of the quad's high part: lanes with quad numbers c4, c4 ^ 1 (same row) are neighbours in the wave
__device__ __forceinline__ void p16_pair_store(char* hi, int c4, unsigned q0h, unsigned q0l, unsigned q1h, unsigned q1l) {
    const bool odd = (c4 & 1) != 0;
    const unsigned rx = __shfl_xor(odd ? q0h : q0l, 1, 64), ry = __shfl_xor(odd ? q1h : q1l, 1, 64);
    const uint4 piece = odd ? make_uint4(rx, ry, q0l, q1l) : make_uint4(q0h, q1h, rx, ry);
    *reinterpret_cast<uint4*>(odd ? hi + 56 : hi) = piece;  // (odd: the low plane's 16 bytes of quads c4 - 1, c4 = 64 bytes on, 8 back)
}
__device__ __forceinline__ float4 p16_load4_rc(const uint2* __restrict__ base, long long row, int cq, int CQ, float inv) {
    const uint2* src = base + row * (2 * CQ) + (cq >> 3) * 16 + (cq & 7);
    const uint2 h = src[0], l = src[8];
    const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
    const f16x2 l0 = __builtin_bit_cast(f16x2, l.x), l1 = __builtin_bit_cast(f16x2, l.y);
    return make_float4(((float)h0.x + (float)l0.x) * inv, ((float)h0.y + (float)l0.y) * inv,
                       ((float)h1.x + (float)l1.x) * inv, ((float)h1.y + (float)l1.y) * inv);
}
__device__ __forceinline__ float4 p16_load4(const uint2* __restrict__ base, long long i, int CQ, float inv) {
    const long long row = i / CQ;
    const int cq = (int)(i - row * CQ);
    const uint2* src = base + row * (2 * CQ) + (cq >> 3) * 16 + (cq & 7);
    const uint2 h = src[0], l = src[8];
    const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
    const f16x2 l0 = __builtin_bit_cast(f16x2, l.x), l1 = __builtin_bit_cast(f16x2, l.y);
    return make_float4(((float)h0.x + (float)l0.x) * inv, ((float)h0.y + (float)l0.y) * inv,
                       ((float)h1.x + (float)l1.x) * inv, ((float)h1.y + (float)l1.y) * inv);
}
// the same tensors as plain bf16 (configs[3]'s arithmetic: the convolutions read bf16 operands; round-to-nearest-even
// here = the rounding the GEMM loader would apply): quad i = 8 bytes at 8 * i, no scale
__device__ __forceinline__ void bf16_store4(uint2* __restrict__ base, long long i, float4 v) {
    base[i] = make_uint2(cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w));
}
__device__ __forceinline__ float4 bf16_load4(const uint2* __restrict__ base, long long i) {
    const uint2 u = base[i];
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
// predicated loads without divergent branches: read from a always-valid address, then select
__device__ __forceinline__ float4 ld4_if(bool ok, const float* __restrict__ p, const float* __restrict__ safe) {
    const float4 v = *reinterpret_cast<const float4*>(ok ? p : safe);
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float ld1_if(bool ok, const float* __restrict__ p, const float* __restrict__ safe) {
    const float v = *(ok ? p : safe);
    return ok ? v : 0.f;
}

}  // namespace trid
