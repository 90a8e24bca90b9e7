// Split-precision GEMM / implicit GEMM, role-alternating ("ping-pong") variant for K-contiguous
// operands (A_KC / A_CONV x B_KC: every forward conv, every data-gradient conv, the linears).
//
// Same arithmetic as gemm_bf16.hip (fp32 operands split on the fly into NPL bf16 planes, NPL=3:
// 6 MFMA products per multiply-add, fp32 accumulate), different schedule.  One 512-thread
// workgroup per CU owns a 256x128 tile; its 8 waves form two groups of four (one wave of each
// group per SIMD).  Group g owns rows [128g, 128g+128) of the tile (64x64 per wave).  The K loop
// runs in half-steps separated by one barrier each:
//
//     half-step 2t   : group 0 issues the 48 MFMAs of K-tile t   | group 1 converts + stores its
//                                                                | share of K-tile t+1 into the
//                                                                | other LDS stage, then issues
//                                                                | its global loads for tile t+2
//     half-step 2t+1 : group 1 computes K-tile t                 | group 0 stages tile t+1 / t+2
//
// so every SIMD always has one wave feeding the matrix pipe while its partner does the VALU
// (bf16 plane split), LDS-store and VMEM work - the two kinds of work overlap by construction
// instead of by luck of workgroup phase.  Global loads stay in flight for a full K-tile (two
// half-steps).  Per-MFMA LDS traffic is 0.55x that of the 128x128 kernel (64x64 per wave).
//
// LDS: 2 stages x NPL planes x (A 256 rows + B 128 rows) x 32 k of bf16 = 145.5 KB (NPL=3).
// Slot map: 16-byte slots (8 consecutive k of one row), slot = kgroup*(R+2) + row: fragment reads
// (16 consecutive rows per ds_read_b128 lane group) and loader writes (8 lanes = 2 rows x 4
// kgroups per ds_write_b128 lane group) are both bank-conflict-free.

#include "split_common.h"

namespace trid {

constexpr int PP_BM = 256, PP_BN = 128, PP_NT = 512;
constexpr int PP_SA = PP_BM + 2, PP_SB = PP_BN + 2;  // kgroup stride, slots
constexpr int PP_PA = 4 * PP_SA, PP_PB = 4 * PP_SB;  // slots per plane image

template <int AMODE, int NPL>
__global__ __launch_bounds__(PP_NT) void gemm_pp_kernel(GemmParams p) {
    constexpr int BM = PP_BM, BN = PP_BN;
    constexpr int STAGE = NPL * (PP_PA + PP_PB);
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
    const int wm = grp * 2 + ((wave >> 1) & 1), wn = wave & 1;  // 4 x 2 waves of 64 x 64
    const int khalf = lane >> 5;

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
#ifdef TRID_PP_TRACE
    // timing probe (tools/pp_trace.py): p.bias carries a u64 buffer [2 groups][64 K-tiles][8 events]
    unsigned long long* trace = (blockIdx.x == 0 && blockIdx.z == 0 && (tid & 255) == 0) ? (unsigned long long*)p.bias : nullptr;
    const float* __restrict__ bias = nullptr;
#define PP_T(kt, ev) do { if (trace && (kt) < 64) trace[(grp * 64 + (kt)) * 8 + (ev)] = __builtin_readcyclecounter(); } while (0)
#else
    const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;
#define PP_T(kt, ev) do { } while (0)
#endif

    // loader lanes of a group (256 threads): kgroup = t&3, row = t>>2 (0..63); the group stages its
    // own 128 A rows (two passes) and half of the B rows.
    //
    // Loads are raw BUFFER loads: per lane a loop-invariant 32-bit byte offset (row start + kgroup),
    // per K-tile one scalar offset (soffset) - no per-tile 64-bit address arithmetic, no branches.
    // Rows beyond M / N have offsets >= num_records and read as zero in hardware; the 3x3 padding
    // taps, 3x3 rows beyond M and a ragged last K-tile invalidate the lane's offset (bit 31) with one select.
    const int lt = tid & 255;
    const int l_kg = lt & 3, l_row = lt >> 2;
    const int a_row0 = grp * 128 + l_row;  // + 64*ps
    const int b_row = grp * 64 + l_row;
    constexpr unsigned OOB = 0x80000000u;  // >= any num_records (tensors < 2 GB, checked at dispatch); +16 cannot wrap

    unsigned voA[2], voB, amask[2];
    const long long a_ld = (AMODE == A_CONV) ? p.Cin : p.lda;
    // A_CONV: the descriptor base sits (W+1) pixels before the tensor so that the per-tap scalar
    // offset ((dy*W+dx) + (W+1)) * Cin stays non-negative; only in-image lanes are ever fetched
    const float* a_base = (AMODE == A_CONV) ? A - (long long)(p.W + 1) * p.Cin : A;
    // (the hardware range check is voffset >= num_records - soffset: the 3x3 descriptor is extended by
    // the largest tap offset and rows >= M are masked out explicitly instead)
    const long long a_rows = (AMODE == A_CONV) ? (long long)p.M + 2 * p.W + 2 : p.M;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (unsigned)(a_rows * a_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (unsigned)((long long)p.N * p.ldb * 4), 0x00020000);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int m = m0 + a_row0 + ps * 64;
        voA[ps] = (unsigned)(((long long)m * a_ld + 8 * l_kg) * 4);
        amask[ps] = 0x1ffu;
        if (AMODE == A_CONV) {
            const uint32_t q = fdiv((uint32_t)m, p.fdW);
            const int x = m - (int)q * p.W;
            const uint32_t b = fdiv(q, p.fdH);
            const int y = (int)q - (int)b * p.H;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << t;
            }
            amask[ps] = m < p.M ? mk : 0u;
        }
    }
    voB = (unsigned)(((long long)(n0 + b_row) * p.ldb + 8 * l_kg) * 4);
    const int ktail = (k_end - k_begin) & (BK - 1);  // A_KC only (9*Cin is a multiple of BK)

    float ra[2][8], rb[8];

    // kt-th K-tile of this split; `valid` false = past the end (nothing is fetched, registers read 0)
    auto load_tiles = [&](int kt, bool valid, bool last) {
        unsigned soA, soB, kill = valid ? 0u : OOB;
        int tap = 0;
        if (AMODE == A_CONV) {  // channel-group-major K order: tile kt = (tap kt%9, channels 32*(kt/9)..)
            tap = kt % 9;
            const int cb = (kt / 9) * BK;
            soA = (unsigned)((((tap / 3) * p.W + (tap % 3)) * p.Cin + cb) * 4);  // (dy+1)*W + (dx+1), see a_base
            soB = (unsigned)((tap * p.Cin + cb) * 4);
        } else {
            soA = soB = (unsigned)((k_begin + kt * BK) * 4);
            if (last && ktail != 0 && 8 * l_kg >= ktail) kill = OOB;
        }
        if (!valid) soA = soB = 0;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            unsigned vo = voA[ps] | kill;
            if (AMODE == A_CONV) vo = ((amask[ps] >> tap) & 1u) ? vo : OOB;
            const float4 u = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, soA, 0));
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo + 16, soA, 0));
            ra[ps][0] = u.x; ra[ps][1] = u.y; ra[ps][2] = u.z; ra[ps][3] = u.w;
            ra[ps][4] = v.x; ra[ps][5] = v.y; ra[ps][6] = v.z; ra[ps][7] = v.w;
        }
        {
            const unsigned vo = voB | kill;
            const float4 u = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, soB, 0));
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, vo + 16, soB, 0));
            rb[0] = u.x; rb[1] = u.y; rb[2] = u.z; rb[3] = u.w; rb[4] = v.x; rb[5] = v.y; rb[6] = v.z; rb[7] = v.w;
        }
    };

    auto store_tiles = [&](uint4* __restrict__ st) {
        uint4* As = st;
        uint4* Bs = st + NPL * PP_PA;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) split_store<NPL, PP_PA>(ra[ps], As + l_kg * PP_SA + a_row0 + ps * 64);
        split_store<NPL, PP_PB>(rb, Bs + l_kg * PP_SB + b_row);
    };

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_slot = wm * 64 + (lane & 31);
    const int b_slot = wn * 64 + (lane & 31);

    auto compute_tile = [&](const uint4* __restrict__ st) {
        const uint4* As = st;
        const uint4* Bs = st + NPL * PP_PA;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kg = 2 * ks + khalf;
            bf16x8 a[NPL][2], b[NPL][2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[pl][i] = __builtin_bit_cast(bf16x8, As[pl * PP_PA + kg * PP_SA + a_slot + 32 * i]);
                    b[pl][i] = __builtin_bit_cast(bf16x8, Bs[pl * PP_PB + kg * PP_SB + b_slot + 32 * i]);
                }
            }
            // smallest terms first (mm, hl, lh, hm, mh, hh); within a term the four MFMAs hit four
            // different accumulators, so none waits on its predecessor
            constexpr int NTERM = (NPL == 3) ? 6 : (NPL == 2) ? 3 : 1;
            constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
            constexpr int TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int t = 6 - NTERM; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t] < NPL ? TA[t] : 0][i], b[TB[t] < NPL ? TB[t] : 0][j],
                                                                            acc[i][j], 0, 0, 0);
        }
    };

    const int T = k_begin < k_end ? (k_end - k_begin + BK - 1) / BK : 0;
    // staging role for K-tile kt+1 (split + LDS store from the registers loaded one K-tile ago), then
    // the global loads of tile kt+2 into the same registers.  The loads are predicated, never
    // branched around, so the registers are one loop-carried value per group.
    auto stage_next = [&](int kt) {
        PP_T(kt, 0);
#ifdef TRID_PP_TRACE
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): separate the wait for the tile's loads from the split work
        PP_T(kt, 6);
#endif
        if (kt + 1 < T) store_tiles(smem4 + ((kt & 1) ^ 1) * STAGE);
        PP_T(kt, 1);
        load_tiles(kt + 2, kt + 2 < T, kt + 2 == T - 1);
        PP_T(kt, 2);
    };
    if (T > 0) {
        load_tiles(0, true, T == 1);
        store_tiles(smem4);
        load_tiles(1, T > 1, T == 2);
        __syncthreads();
        // two straight-line loops (same barrier count), one per group: no role-dependent value merges
        if (grp == 0) {
            for (int kt = 0; kt < T; ++kt) {
                PP_T(kt, 4);
                compute_tile(smem4 + (kt & 1) * STAGE);
                PP_T(kt, 5);
                __syncthreads();
                stage_next(kt);
                __syncthreads();
                PP_T(kt, 3);
            }
        } else {
            for (int kt = 0; kt < T; ++kt) {
                stage_next(kt);
                __syncthreads();
                PP_T(kt, 3);
                PP_T(kt, 4);
                compute_tile(smem4 + (kt & 1) * STAGE);
                PP_T(kt, 5);
                __syncthreads();
            }
        }
    }

    // ---- epilogue (same contract as gemm.hip / gemm_bf16.hip) ---------------------------
    const int row_base = m0 + wm * 64 + 4 * khalf;
    const int col_base = n0 + wn * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col_base + 32 * j;
        const float bv = (bias != nullptr && col < p.N) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
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
                float v = p.alpha * acc[i][j][r] + bv;
                if (p.accumulate) v += oldv[r];
                if (row < p.M && col < p.N) C[(long long)row * p.ldc + col] = v;
                acc[i][j][r] = v;
            }
        }
    }

    if (p.stats != nullptr) {
        // BatchNorm partials per 128-row group = per wave group: (mean, M2) of every column
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem4);  // [4 wave rows][BN]
        const int rows_left = p.M - (m0 + grp * 128);
        const int cnt = rows_left < 128 ? (rows_left > 0 ? rows_left : 1) : 128;
        const float inv = 1.f / (float)cnt;
        float mean[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) s += acc[i][j][r];
                }
            s += __shfl_xor(s, 32, 64);
            if (khalf == 0) red[wm * BN + wn * 64 + 32 * j + (lane & 31)] = s;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cl = wn * 64 + 32 * j + (lane & 31);
            mean[j] = (red[(2 * grp) * BN + cl] + red[(2 * grp + 1) * BN + cl]) * inv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                    const float d = acc[i][j][r] - mean[j];
                    if (row < p.M) s += d * d;
                }
            s += __shfl_xor(s, 32, 64);
            if (khalf == 0) red[wm * BN + wn * 64 + 32 * j + (lane & 31)] = s;
        }
        __syncthreads();
        if ((wm & 1) == 0 && khalf == 0 && rows_left > 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + 32 * j + (lane & 31);
                const int col = n0 + cl;
                if (col < p.N) {
                    float* dst = p.stats + ((long long)(mb * 2 + grp) * p.N + col) * 2;
                    dst[0] = mean[j];
                    dst[1] = red[(2 * grp) * BN + cl] + red[(2 * grp + 1) * BN + cl];
                }
            }
        }
    }
}

template <int AMODE, int NPL>
static int launch_pp(GemmParams& p, hipStream_t stream) {
    p.mblocks = (p.M + PP_BM - 1) / PP_BM;
    p.nblocks = (p.N + PP_BN - 1) / PP_BN;
    dim3 grid((unsigned)(p.mblocks * p.nblocks), 1, (unsigned)(p.batch * p.splits));
    constexpr size_t lds = (size_t)2 * NPL * (PP_PA + PP_PB) * sizeof(uint4);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pp_kernel<AMODE, NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("trid_gemm_f32(pp): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_pp_kernel<AMODE, NPL>), grid, dim3(PP_NT), lds, stream, p);
    return check_launch("trid_gemm_f32(pp)");
}

int gemm_pp_dispatch(GemmParams& p, int a_mode, int b_mode, int precision, hipStream_t stream) {
    if (b_mode != B_KC || (a_mode != A_KC && a_mode != A_CONV)) return TRID_E_UNSUPPORTED;
    if (precision == 6) return a_mode == A_KC ? launch_pp<A_KC, 3>(p, stream) : launch_pp<A_CONV, 3>(p, stream);
    if (precision == 1) return a_mode == A_KC ? launch_pp<A_KC, 1>(p, stream) : launch_pp<A_CONV, 1>(p, stream);
    return TRID_E_UNSUPPORTED;
}

}  // namespace trid
