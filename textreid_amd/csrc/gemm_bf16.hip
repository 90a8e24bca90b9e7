// Split-precision GEMM / implicit-GEMM: fp32 operands, fp32 accumulate, bf16 MFMA.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA.  Each fp32
// operand is split ON THE FLY (while being staged into LDS) into NPL bf16 planes
//     x = hi + mid (+ lo),  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// and a product a*b is evaluated as the sum of the significant plane products:
//     NPL = 3 ("bf16x6"): hh + hm + mh + hl + lh + mm   (dropped terms <= 2^-26 |ab|: fp32-class)
//     NPL = 2 ("bf16x3"): hh + hm + mh                  (dropped terms ~ 2^-17 |ab|)
//     NPL = 1 ("bf16")  : hh                            (operands rounded to bf16, fp32 accumulate:
//                                                        the arithmetic of a bf16 autocast run)
// all accumulated in the MFMA's fp32 accumulator, small terms first.  Same loader modes,
// epilogues (alpha / bias / accumulate / split-K / BatchNorm partials) and C ABI as gemm.hip.
//
// Tile 128x128x32, 8 waves (2x4, 64x32 per wave), single LDS stage + register prefetch;
// 2 workgroups per CU.  LDS image: 16-byte slots holding 8 consecutive k of one
// row, slot = kgroup*(128+1) + row, one image per plane -> every MFMA operand fetch is one
// conflict-free ds_read_b128 per plane.

#include <mutex>

#include "split_common.h"

namespace trid {

constexpr int BM_T = 128;             // tile height
constexpr int NT = 512;               // 8 waves
#ifndef TRID_F16_WAVES_KC
#define TRID_F16_WAVES_KC 6  // ... three where both operands are K-contiguous (fwd / 3x3 dgrad: 76-80 VGPRs, 37 KB of LDS each)
#endif
#ifndef TRID_F16_WAVES
#define TRID_F16_WAVES 4  // fp16-split kernels: two 8-wave workgroups per CU (<= 128 VGPRs)
#endif
#ifndef TRID_NARROW_WAVES
#define TRID_NARROW_WAVES 6  // 64-column K-contiguous tiles: 40 KB of LDS -> 3 workgroups per CU if the kernel fits 80 VGPRs
#endif
// Tile 128 x BN x 32, one LDS stage + register prefetch, 2 workgroups per CU.
//   BN = 128: 2 x 4 waves, 64 x 32 per wave;   BN = 64 (outputs with <= 64 columns: stem, layer1):
//   4 x 2 waves, 32 x 32 per wave - half the MFMA work per tile for the same A staging, still ahead
//   of the fp32-input MFMA kernel whose peak is 16x lower.
// B image: the swizzled wide layout needs 576 slots; a K-contiguous 64-row image only 4 x 65
__host__ __device__ constexpr int b_plane_slots(int bmode, int bn) { return (bmode == B_KC && bn <= 64) ? 4 * (bn + TRID_KC_PAD) : plane_slots(bn); }
// BN = 32 (the 32-channel stem convolutions): 256 x 32 tiles, 8 x 1 waves of 32 x 32 - on the 128 x 64 tile half of the
// waves had no columns to work on
__host__ __device__ constexpr int tile_rows(int bn) { return bn == 32 ? 256 : BM_T; }

// ARITH: 1 / 2 / 3 = number of bf16 planes (1, 3, 6 products); 16 = two fp16 planes, 3 products, one accumulator
template <int AMODE, int BMODE, int ARITH, int BN>
__global__ __launch_bounds__(NT, (BN == 64 && BMODE == B_KC && TRID_NARROW_WAVES > 0 && ARITH != 16) ? TRID_NARROW_WAVES : (ARITH == 16 ? ((BMODE == B_KC && AMODE != A_MC && BN != 32) ? TRID_F16_WAVES_KC : TRID_F16_WAVES) : 1)) void gemm_bf16s_kernel(GemmParams p) {
    constexpr bool F16 = (ARITH == 16);
    constexpr int NPL = F16 ? 2 : ARITH;
    constexpr int BM = tile_rows(BN);
    constexpr int WAVES_N = BN / 32;           // 4 or 2
    constexpr int WAVES_M = 8 / WAVES_N;       // 2 or 4
    constexpr int TM = BM / (32 * WAVES_M);    // 32-row MFMA tiles per wave: 2 or 1
    constexpr int PA = plane_slots(BM), PB = b_plane_slots(BMODE, BN);
    // M/N-contiguous operands are loaded WIDE (float4 along the rows, 4 consecutive k per lane) and
    // transposed through registers into the swizzled LDS image
    constexpr bool A_WIDE = (AMODE == A_MC);
    constexpr bool B_WIDE = (BMODE != B_KC);
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

    if (p.gate != nullptr && *p.gate == 0) return;  // predicated launch (retrieval overflow fallback)
    // f16x3: per-tensor power-of-two scales from the operands' largest magnitudes (device scalars)
    float scaleA = 1.f, scaleB = 1.f;
    if (F16) {
        if (p.a_amax != nullptr) scaleA = f16_scale_of(*p.a_amax);
        if (p.b_amax != nullptr) scaleB = f16_scale_of(*p.b_amax);
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int khalf = lane >> 5;
    // the second-dispatched half of the workgroup loses the age-based issue arbitration on every
    // segment (its 24 MFMAs take 2000 cycles against 1150 for waves 0-3 in a cycle trace): static priority
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    const uint32_t nwg = (uint32_t)p.mblocks * (uint32_t)p.nblocks;
    const uint32_t lid = xcd_remap(blockIdx.x, nwg);
    const int mb = lid / p.nblocks, nb = lid % p.nblocks;
    const int m0 = mb * BM, n0 = nb * BN;
    const int z = blockIdx.z;
    const int bz = z / p.splits, sz = z % p.splits;
    const int k_begin = sz * p.k_chunk;
    const int k_end = min(p.K, k_begin + p.k_chunk);

    const float* __restrict__ A = p.A + (long long)bz * p.sA;
    const float* __restrict__ Bp = p.B + (long long)bz * p.sB;
    float* __restrict__ C = p.C + (long long)bz * p.sC + (long long)sz * p.sSplit;
    const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;

    // loader lanes.  K-contiguous operands: all 512 threads, (kg = tid&3, row = tid>>2), one 8-k slot each.
    // Wide (M/N-contiguous) operands: 256 threads per operand (A: waves 0-3, B: waves 4-7),
    // (mq = t&31 -> rows 4mq..4mq+3, kq = t>>5 -> k 4kq..4kq+3), four float4 loads each.
    //
    // Loads are raw BUFFER loads wherever the address splits into a loop-invariant per-lane byte offset
    // and a per-tile scalar offset (everything except the B_CONV gather and 3x3 inputs whose channel
    // count is not a multiple of the K-tile): no per-tile 64-bit address arithmetic, no branches around
    // the loads.  Out-of-range rows read as zero in hardware (voffset >= num_records - soffset); padding
    // taps, a ragged last K-tile and lanes beyond M/N set bit 31 of the offset instead.
    constexpr bool A_K = (AMODE != A_MC);
    const int a_kg = A_K ? (tid & 3) : (tid >> 7), a_row = A_K ? (tid >> 2) : (tid & 127);
    const int b_kg = tid & 3, b_row = tid >> 2;  // B_KC only
    // wide lanes: (mq, kq) with kq's low bit (the 8-byte half of a slot) in the lane's low bit, so that two neighbouring
    // lanes fill one 16-byte slot and a 16-lane ds_write_b64 group writes 128 contiguous bytes (with mq in the low
    // bits the group strode 16 bytes and used half of the banks: 29 % of the LDS-active cycles were conflicts)
    const int w_mq = (tid >> 1) & 31, w_kq = 2 * ((tid & 255) >> 6) + (tid & 1);
    const bool a_wide_lane = tid < 256, b_wide_lane = tid >= 256;
    constexpr unsigned OOB = 0x80000000u;
    // channel-group-major K order of the 3x3 implicit GEMM (see tile_k below): tap is uniform per tile
    const bool permute = (AMODE == A_CONV) && (BMODE == B_KC) && (p.Cin % BK == 0) && p.splits == 1;

    const long long a_ld = (AMODE == A_CONV) ? p.Cin : p.lda;
    const long long a_rows = (AMODE == A_KC) ? p.M : (AMODE == A_MC) ? p.K : (long long)p.M + 2 * p.W + 2;
    const float* a_base = (AMODE == A_CONV) ? A - (long long)(p.W + 1) * p.Cin : A;  // tap offsets stay >= 0
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (unsigned)(a_rows * a_ld * 4), 0x00020000);
    const long long b_rows = (BMODE == B_KC) ? p.N : p.K;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (unsigned)(b_rows * p.ldb * 4), 0x00020000);

    constexpr int APASS = BM / 128;  // a loader pass covers 128 A rows
    int a_y[APASS], a_x[APASS];
    unsigned voA[APASS], amask[APASS];
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
        const int m = m0 + a_row + ps * 128;
        voA[ps] = (unsigned)(((long long)m * a_ld + 8 * a_kg) * 4);
        amask[ps] = 0x1ffu;
        if (AMODE == A_CONV) {
            const uint32_t q = fdiv((uint32_t)m, p.fdW);
            a_x[ps] = m - (int)q * p.W;
            const uint32_t b = fdiv(q, p.fdH);
            a_y[ps] = (int)q - (int)b * p.H;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = a_y[ps] + t / 3 - 1, xx = a_x[ps] + t % 3 - 1;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << t;
            }
            amask[ps] = m < p.M ? mk : 0u;
        }
    }
    // wide lanes: rows are k, lanes run along m / n; a lane beyond M / N is dead for the whole kernel
    const unsigned voAw = (m0 + 4 * w_mq < p.M) ? (unsigned)(((long long)(4 * w_kq) * p.lda + m0 + 4 * w_mq) * 4) : OOB;
    const unsigned voBw = (4 * w_mq < BN && n0 + 4 * w_mq < p.N) ? (unsigned)(((long long)(4 * w_kq) * p.ldb + n0 + 4 * w_mq) * 4) : OOB;
    const unsigned voB = b_row < BN ? (unsigned)(((long long)(n0 + b_row) * p.ldb + 8 * b_kg) * 4) : OOB;
    int b_dy = 0, b_dx = 0, b_c = 0;
    if (BMODE == B_CONV) {
        const int j = n0 + 4 * w_mq;
        const uint32_t tap = fdiv((uint32_t)j, p.fdC);
        b_c = j - (int)tap * p.Cin;
        b_dy = (int)tap / 3 - 1;
        b_dx = (int)tap % 3 - 1;
    }

    float ra[APASS][8], rb[8];
    float4 wa[4], wb[4];

    auto ldb4 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned vo, unsigned so) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
    };

    // k0: K offset of the tile (already permuted for the 3x3 fast path)
    auto load_tiles = [&](int k0) {
        // ---- A ----
        if (AMODE == A_KC || (AMODE == A_CONV && permute)) {
            unsigned so, kill = 0;
            int tap = 0;
            if (AMODE == A_CONV) {
                tap = k0 / p.Cin;  // uniform: a tile never straddles taps when Cin % BK == 0
                so = (unsigned)((((tap / 3) * p.W + (tap % 3)) * p.Cin + (k0 - tap * p.Cin)) * 4);
            } else {
                so = (unsigned)(k0 * 4);
                kill = (k0 + 8 * a_kg < k_end) ? 0u : OOB;
            }
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) {
                unsigned vo = voA[ps] | kill;
                if (AMODE == A_CONV) vo = ((amask[ps] >> tap) & 1u) ? vo : OOB;
                const float4 u = ldb4(rsA, vo, so), v = ldb4(rsA, vo + 16, so);
                ra[ps][0] = u.x; ra[ps][1] = u.y; ra[ps][2] = u.z; ra[ps][3] = u.w;
                ra[ps][4] = v.x; ra[ps][5] = v.y; ra[ps][6] = v.z; ra[ps][7] = v.w;
            }
        } else if (AMODE == A_CONV) {  // general 3x3 gather (channel counts that are not a multiple of the K-tile)
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) {
                const int m = m0 + a_row + ps * 128;
                const int k = k0 + 8 * a_kg;
                const uint32_t tap = fdiv((uint32_t)k, p.fdC);
                const int c = k - (int)tap * p.Cin;
                const int dy = (int)tap / 3 - 1, dx = (int)tap % 3 - 1;
                const int yy = a_y[ps] + dy, xx = a_x[ps] + dx;
                const bool ok = m < p.M && k < k_end && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const float* src = A + (long long)(m + dy * p.W + dx) * p.Cin + c;
                const float4 u = ld4_if(ok, src, A), v = ld4_if(ok, src + 4, A);
                ra[ps][0] = u.x; ra[ps][1] = u.y; ra[ps][2] = u.z; ra[ps][3] = u.w;
                ra[ps][4] = v.x; ra[ps][5] = v.y; ra[ps][6] = v.z; ra[ps][7] = v.w;
            }
        }
        if (A_WIDE && a_wide_lane) {  // A_MC wide: float4 along m for 4 consecutive k (rows k >= K read as zero)
#pragma unroll
            for (int j = 0; j < 4; ++j) wa[j] = ldb4(rsA, voAw, (unsigned)((long long)(k0 + j) * p.lda * 4));
        }
        // ---- B ----
        if (BMODE == B_KC) {
            const unsigned vo = voB | ((k0 + 8 * b_kg < k_end) ? 0u : OOB);
            const unsigned so = (unsigned)(k0 * 4);
            const float4 u = ldb4(rsB, vo, so), v = ldb4(rsB, vo + 16, so);
            rb[0] = u.x; rb[1] = u.y; rb[2] = u.z; rb[3] = u.w; rb[4] = v.x; rb[5] = v.y; rb[6] = v.z; rb[7] = v.w;
        } else if (BMODE == B_NC) {
            if (b_wide_lane) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wb[j] = ldb4(rsB, voBw, (unsigned)((long long)(k0 + j) * p.ldb * 4));
            }
        } else {  // B_CONV: row k is a pixel, 4 consecutive columns n = (tap, c..c+3)
            if (b_wide_lane) {
                const int n = n0 + 4 * w_mq;
                const int kbase = k0 + 4 * w_kq;  // multiple of 4
                if ((p.W & 3) == 0) {
                    // 4 consecutive pixels stay in one image row when W % 4 == 0: one decomposition
                    const uint32_t q = fdiv((uint32_t)kbase, p.fdW);
                    const int x0 = kbase - (int)q * p.W;
                    const uint32_t b = fdiv(q, p.fdH);
                    const int yy = (int)q - (int)b * p.H + b_dy;
                    const bool row_ok = 4 * w_mq < BN && n < p.N && yy >= 0 && yy < p.H;
                    const float* src = Bp + (long long)(kbase + b_dy * p.W + b_dx) * p.Cin + b_c;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int xx = x0 + j + b_dx;
                        wb[j] = ld4_if(row_ok && (kbase + j) < k_end && xx >= 0 && xx < p.W, src + (long long)j * p.Cin, Bp);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = kbase + j;
                        const uint32_t q = fdiv((uint32_t)k, p.fdW);
                        const int x = k - (int)q * p.W;
                        const uint32_t b = fdiv(q, p.fdH);
                        const int y = (int)q - (int)b * p.H;
                        const int yy = y + b_dy, xx = x + b_dx;
                        wb[j] = ld4_if(4 * w_mq < BN && n < p.N && k < k_end && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W,
                                       Bp + (long long)(k + b_dy * p.W + b_dx) * p.Cin + b_c, Bp);
                    }
                }
            }
        }
    };

    auto store_tiles = [&](uint4* __restrict__ As, uint4* __restrict__ Bs) {
        if (F16) {
            if (A_WIDE) {
                if (a_wide_lane) split_store_wide_f16<PA>(wa, scaleA, As, w_mq, w_kq);
            } else {
#pragma unroll
                for (int ps = 0; ps < APASS; ++ps) split_store_f16<PA>(ra[ps], scaleA, As + slot_of<true, BM>(a_kg, a_row + 128 * ps));
            }
            if (B_WIDE) {
                if (b_wide_lane && 4 * w_mq < BN) split_store_wide_f16<PB>(wb, scaleB, Bs, w_mq, w_kq);
            } else if (b_row < BN) {
                split_store_f16<PB>(rb, scaleB, Bs + slot_of<true, BN>(b_kg, b_row));
            }
            return;
        }
        if (A_WIDE) {
            if (a_wide_lane) split_store_wide<NPL, PA>(wa, As, w_mq, w_kq);
        } else {
#pragma unroll
            for (int ps = 0; ps < APASS; ++ps) split_store<NPL, PA>(ra[ps], As + slot_of<true, BM>(a_kg, a_row + 128 * ps));
        }
        if (B_WIDE) {
            if (b_wide_lane && 4 * w_mq < BN) split_store_wide<NPL, PB>(wb, Bs, w_mq, w_kq);
        } else if (b_row < BN) {
            split_store<NPL, PB>(rb, Bs + slot_of<true, BN>(b_kg, b_row));
        }
    };

    v16f acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int a_slot = wm * (32 * TM) + (lane & 31);
    const int b_slot = wn * 32 + (lane & 31);

    auto compute_step = [&](const uint4* __restrict__ As, const uint4* __restrict__ Bs, int ks) {
        if constexpr (F16) {
            f16x8 a[2][TM], b[2];
            const int kg = 2 * ks + khalf;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[pl][i] = __builtin_bit_cast(f16x8, As[pl * PA + slot_of<!A_WIDE, BM>(kg, a_slot + 32 * i)]);
                b[pl] = __builtin_bit_cast(f16x8, Bs[pl * PB + slot_of<!B_WIDE, BN>(kg, b_slot)]);
            }
            // small terms first; consecutive MFMAs hit different accumulators where the wave has two
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0], acc[i], 0, 0, 0);
            return;
        }
        bf16x8 a[NPL][TM], b[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const int kg = 2 * ks + khalf;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[pl][i] = __builtin_bit_cast(bf16x8, As[pl * PA + slot_of<!A_WIDE, BM>(kg, a_slot + 32 * i)]);
            b[pl] = __builtin_bit_cast(bf16x8, Bs[pl * PB + slot_of<!B_WIDE, BN>(kg, b_slot)]);
        }
        // Term-major order: consecutive MFMAs hit DIFFERENT accumulators where the wave has two, so no
        // MFMA waits on the result of the one issued just before it.  Smallest terms first: mm, hl, lh
        // (3 planes only), then hm, mh, hh.
        constexpr int NTERM = (NPL == 3) ? 6 : (NPL == 2) ? 3 : 1;
        constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
        constexpr int TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int t = 6 - NTERM; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t] < NPL ? TA[t] : 0][i], b[TB[t] < NPL ? TB[t] : 0], acc[i], 0, 0, 0);
    };

    // 3x3 implicit GEMM: walk K channel-group-major (all 9 taps of a 32-channel slab back to
    // back) instead of tap-major, so the nine shifted re-reads of one activation slab are adjacent
    // in time and hit L1/L2 instead of going back to the Infinity Cache / HBM.
    auto tile_k = [&](int k0) {
        if (!permute) return k0;
        const int kt = k0 / BK;
        return (kt % 9) * p.Cin + (kt / 9) * BK;
    };

    if (k_begin < k_end) {
        load_tiles(tile_k(k_begin));
        store_tiles(smem4, smem4 + NPL * PA);
        __syncthreads();
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            const bool more = (k0 + BK) < k_end;
            if (more) load_tiles(tile_k(k0 + BK));
            const uint4* As = smem4;
            const uint4* Bs = As + NPL * PA;
            compute_step(As, Bs, 0);
            compute_step(As, Bs, 1);
            __syncthreads();
            if (more) {
                store_tiles(smem4, smem4 + NPL * PA);
                __syncthreads();
            }
        }
    }

    // ---- epilogue (same contract as gemm.hip) -----------------------------------------
    if (F16) {  // undo the operand scales (exact: powers of two)
        const float unscale = 1.f / (scaleA * scaleB);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] *= unscale;
    }
    const int row_base = m0 + wm * (32 * TM) + 4 * khalf;
    const int col = n0 + wn * 32 + (lane & 31);
    if (p.filt.thr != nullptr) {
        // top-k admission filter: nothing is stored but the (rare) elements that reach the row's threshold
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float thr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                thr[r] = (row < p.M && col < p.N) ? p.filt.thr[(long long)row * p.filt.thr_stride] : INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = p.alpha * acc[i][r];
                if (v >= thr[r]) {
                    const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                    const int slot = atomicAdd(p.filt.cnt + row, 1);
                    if (slot < p.filt.cap) {
                        float2* dst = reinterpret_cast<float2*>(p.filt.cand) + (long long)row * p.filt.cap + slot;
                        *dst = make_float2(v, __int_as_float(col + p.filt.col0));
                    } else {
                        *p.filt.overflow = 1;
                    }
                }
            }
        }
        return;
    }
    const float bv = (bias != nullptr && col < p.N) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float oldv[16];
        if (p.accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                oldv[r] = (row < p.M && col < p.N) ? C[(long long)row * p.ldc + col] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
            float v = p.alpha * acc[i][r] + bv;
            if (p.accumulate) v += oldv[r];
            if (p.res != nullptr && row < p.M && col < p.N) v += p.res[(long long)row * p.ldres + col];
            if (p.relu) v = fmaxf(v, 0.f);
            if (row < p.M && col < p.N) C[(long long)row * p.ldc + col] = v;
            acc[i][r] = v;
        }
    }

    if (p.stats != nullptr) {
        // BatchNorm partials per 128-row slab of the tile (one slab, or two for the 256-row tile): column (mean, M2)
        // over the slab's rows < M
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem4);  // [WAVES_M][BN]
        constexpr int WPS = 128 / (32 * TM);            // waves (along M) per slab
        const int slab = wm / WPS;
        const int rows_left = p.M - (m0 + 128 * slab);
        const int cnt = rows_left < 128 ? rows_left : 128;
        const float inv = cnt > 0 ? 1.f / (float)cnt : 0.f;
        const int cl = wn * 32 + (lane & 31);
        auto column_total = [&](float s) {  // sum over the slab's 128 rows of a per-lane partial
            s += __shfl_xor(s, 32, 64);
            if (khalf == 0) red[wm * BN + cl] = s;
            __syncthreads();
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WPS; ++w) t += red[(slab * WPS + w) * BN + cl];
            __syncthreads();
            return t;
        };
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.M) s += acc[i][r];
            }
        const float mean = column_total(s) * inv;
        s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                const float d = acc[i][r] - mean;
                if (row < p.M) s += d * d;
            }
        const float m2 = column_total(s);
        if ((wm % WPS) == 0 && khalf == 0 && col < p.N && cnt > 0) {
            float* dst = p.stats + (((long long)mb * (BM / 128) + slab) * p.N + col) * 2;
            dst[0] = mean;
            dst[1] = m2;
        }
    }
}

template <int AMODE, int BMODE, int ARITH, int BN>
static int launch_bf16(GemmParams& p, hipStream_t stream) {
    constexpr int NPL = ARITH == 16 ? 2 : ARITH;
    p.mblocks = (p.M + tile_rows(BN) - 1) / tile_rows(BN);
    p.nblocks = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.mblocks * p.nblocks), 1, (unsigned)(p.batch * p.splits));
    constexpr size_t lds = (size_t)NPL * (plane_slots(tile_rows(BN)) + b_plane_slots(BMODE, BN)) * sizeof(uint4);
    // once per kernel instantiation and process, safe under concurrent first calls from several host threads
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        if (lds > 48 * 1024)
            attr_err = hipFuncSetAttribute((const void*)gemm_bf16s_kernel<AMODE, BMODE, ARITH, BN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_gemm_f32(split): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((gemm_bf16s_kernel<AMODE, BMODE, ARITH, BN>), grid, dim3(NT), lds, stream, p);
    return check_launch("trid_gemm_f32(split)");
}

template <int AMODE, int BMODE, int NPL>
static int pick_tile(GemmParams& p, hipStream_t stream) {
    // 128-row tiles, two workgroups per CU; 64 columns when the output is that narrow.  (A 256x128
    // double-buffered tile with one workgroup per CU and a role-alternating 256x128 schedule were both
    // measured 3-4 % slower on the same box)
    if constexpr (NPL == 16 && BMODE == B_KC && AMODE != A_MC) {
        if (p.N <= 32) return launch_bf16<AMODE, BMODE, NPL, 32>(p, stream);  // stem convolutions
    }
    if (p.N <= 64) return launch_bf16<AMODE, BMODE, NPL, 64>(p, stream);
    return launch_bf16<AMODE, BMODE, NPL, 128>(p, stream);
}

template <int NPL>
static int dispatch_modes(GemmParams& p, int am, int bm, hipStream_t stream) {
    if (am == A_KC && bm == B_KC) return pick_tile<A_KC, B_KC, NPL>(p, stream);
    if (am == A_CONV && bm == B_KC) return pick_tile<A_CONV, B_KC, NPL>(p, stream);
    if (am == A_KC && bm == B_NC) return pick_tile<A_KC, B_NC, NPL>(p, stream);
    if (am == A_MC && bm == B_NC) return pick_tile<A_MC, B_NC, NPL>(p, stream);
    if (am == A_MC && bm == B_CONV) return pick_tile<A_MC, B_CONV, NPL>(p, stream);
    return TRID_E_UNSUPPORTED;
}

int gemm_bf16_dispatch(GemmParams& p, int a_mode, int b_mode, int precision, hipStream_t stream) {
    if (precision == 16) return dispatch_modes<16>(p, a_mode, b_mode, stream);
    if (precision == 6) return dispatch_modes<3>(p, a_mode, b_mode, stream);
    if (precision == 1) return dispatch_modes<1>(p, a_mode, b_mode, stream);
    return dispatch_modes<2>(p, a_mode, b_mode, stream);
}

}  // namespace trid
