// GEMM / implicit-GEMM on PRE-SPLIT operands ("P16" tensors), LDS-DMA staged.
//
// The fp16 two-plane arithmetic of gemm_bf16.hip (x' = x * 2^s, hi = fp16(x'), lo = fp16(x' - hi), product =
// hi*hi + hi*lo + lo*hi in one fp32 MFMA accumulator: <= 3 * 2^-22 per product, fp32-class) with the split moved
// OUT of the GEMM: the kernel that PRODUCES an operand (BatchNorm apply / backward, the per-step weight pass) writes
// it already split, once, instead of every GEMM tile re-splitting every element it stages (nine times per
// activation for a 3x3 convolution, once per column tile on top).
//
// P16 layout of a [R rows][K] operand (K % 32 == 0), 4 bytes per element like fp32:
//     row r, 32-wide K group g:  128 bytes at r*K*4 + g*128 = [ hi(k = 32g .. 32g+31) : 64 B | lo(same k) : 64 B ]
// so one row of one 32-deep K tile is ONE full 128-byte line, and a loader lane moves 16 bytes = 8 consecutive k of
// one plane = exactly one MFMA 32x32x16 operand fragment.  Loaders are pure copies: `buffer_load_dwordx4 ... lds`
// (LDS-DMA, no VGPR staging, no VALU, no ds_write), eight lanes per 128-byte row, 8 rows per wave instruction.
//
// LDS image of a tile: [rows][8 slots of 16 B], slot s of row r stored at position s ^ ((r >> 1) & 7).  LDS-DMA
// writes lane-linear (wave base + 16*lane), so the permutation is applied to the SOURCE address of each lane; the
// MFMA fragment reads (32 consecutive rows, one slot, ds_read_b128 in 16-lane groups {0-3,12-15,20-27} / {4-11,
// 16-19,28-31}) then hit 16 distinct 16-byte bank groups: conflict-free.
//
// Pipeline: STAGES LDS buffers, DMA issued STAGES-1 tiles ahead, ONE barrier per K tile, counted s_waitcnt vmcnt
// (never 0 in steady state for STAGES >= 3), raw s_barrier so in-flight DMA survives the barrier.

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "split_common.h"

namespace trid {

constexpr int P16_BK = 32;

// LDS of a gemm_p16_kernel workgroup: the operand stages - and, for the 12-wave tile that has a CU to itself anyway, room for the
// whole fp32 tile of the staged (whole-row) epilogue
constexpr int p16_lds_bytes(int BM, int BN, int NW, int STAGES) {
    const int st = STAGES * (BM + BN) * 128;
    return (NW == 12 && BM * BN * 4 > st) ? BM * BN * 4 : st;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one LDS-DMA instruction: every lane copies 16 bytes from its own source offset to (wave-uniform LDS base) + 16 * lane
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& rs, uint4* lds_base, unsigned voffset, unsigned soffset) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_base, 16, voffset, soffset, 0, 0);
}

// The same instruction out of the compiler's sight: hipcc waits `vmcnt(0)` in front of every LDS read that follows an
// LDS-DMA it knows about (it cannot tell the DMA's stage from the read's), which would drain the prefetch in the middle of
// the software-pipelined loop.  Completion is counted by hand there (s_waitcnt vmcnt + s_barrier at the top of a tile).
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i_t raw_rsrc(const void* base, unsigned bytes) {  // the descriptor make_buffer_rsrc builds, as four provably uniform words
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    v4i_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16_asm(const v4i_t& rs, unsigned lds_byte_addr, unsigned voffset, unsigned soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)), "v"(voffset), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(soffset)) : "memory", "m0");
}

// AMODE: A_KC (rows = GEMM rows) or A_CONV (rows = pixels of an NHWC image, K = 9 taps x Cin, 3x3 / stride 1 / pad 1)
// PL = 2: P16 operands (two fp16 planes, 32 k per 128-byte row chunk, 3 MFMA products per multiply-add: fp32-class);
// PL = 1: plain bf16 operands (64 k per 128-byte chunk, one bf16 MFMA per product): configs[3]'s bf16 arithmetic on
// tensors their producers already wrote in bf16 - half the operand bytes, a third of the matrix work.
// PP: the two wave halves of the workgroup run the K loop half a tile apart ("ping-pong", see the main loop).
// SP: software-pipelined main loop (see "software pipeline" below): the fragments of a K tile's second half are
// multiplied AFTER the next tile's barrier, under the LDS reads of that tile's first half.
template <int AMODE, int BM, int BN, int WM, int WN, int STAGES, int PL, bool PP = false, int SP = 0>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN == 4 && STAGES == 2) ? 2 : (WM * WN == 8 && (BM + BN) * 128 * STAGES <= 80 * 1024 ? 4 : ((WM * WN == 6 || WM * WN == 12) ? 3 : 2))) void gemm_p16_kernel(GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int BKE = PL == 2 ? 32 : 64;  // K elements per 128-byte row chunk = per K tile
    constexpr int EB = PL == 2 ? 4 : 2;     // bytes per element of a row
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int A_CH = BM / 8, B_CH = BN / 8;                          // 1-KB DMA chunks (8 rows x 128 B)
    constexpr int A_PW = (A_CH + NW - 1) / NW, B_PW = (B_CH + NW - 1) / NW;  // chunks per wave
    constexpr int PER_TILE = A_PW + B_PW;                                // DMA instructions per wave and K tile
    constexpr int STAGE_SLOTS = (BM + BN) * 8;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];

    const float scaleA = (PL == 2 && p.a_amax != nullptr) ? f16_scale_of(*p.a_amax) : 1.f;
    const float scaleB = (PL == 2 && p.b_amax != nullptr) ? f16_scale_of(*p.b_amax) : 1.f;
    // eval epilogue (c_fmt 1): its device scalars are fetched here, under the K loop, not at the tile's tail
    float ev_bound = 0.f, ev_rinv = 1.f;
    unsigned ev_seen = 0;
    if (p.c_fmt == 1) {
        ev_bound = eval_out_bound(p.ev);
        ev_rinv = p.res16 != nullptr ? 1.f / f16_scale_of(*p.res16_amax) : 1.f;
        // what earlier workgroups folded so far (device-scope load: the atomics live beyond the per-XCD L2)
        if (p.ev.out_tmax != nullptr) ev_seen = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.ev.out_tmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int khalf = lane >> 5;

    const uint32_t nwg = (uint32_t)p.mblocks * (uint32_t)p.nblocks;
    const uint32_t lid = xcd_remap(blockIdx.x, nwg);
    const int mb = lid / p.nblocks, nb = lid % p.nblocks;
    const int m0 = mb * BM, n0 = nb * BN;
    const int z = blockIdx.z;
    const int bz = z / p.splits, sz = z % p.splits;
    const int kt_begin = sz * (p.k_chunk / BKE);
    const int kt_end = min(p.K / BKE, kt_begin + p.k_chunk / BKE);
    const int nk = kt_end - kt_begin;

    const char* A = reinterpret_cast<const char*>(p.A) + (long long)bz * p.sA * EB;
    const char* Bp = reinterpret_cast<const char*>(p.B) + (long long)bz * p.sB * EB;
    float* __restrict__ C = p.C + (long long)bz * p.sC + (long long)sz * p.sSplit;
    const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;

    // ---- loader lanes: chunk c covers tile rows 8c .. 8c+7; lane -> (row 8c + lane/8, stored slot lane%8)
    constexpr unsigned OOB = 0x80000000u;
    const long long a_ld_bytes = (AMODE == A_CONV ? (long long)p.Cin : p.lda) * EB;
    const long long a_rows = (AMODE == A_CONV) ? (long long)p.M + 2 * p.W + 2 : p.M;
    const char* a_base = (AMODE == A_CONV) ? A - (long long)(p.W + 1) * a_ld_bytes : A;  // tap offsets stay >= 0
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (unsigned)(a_rows * a_ld_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (unsigned)((long long)p.N * p.ldb * EB), 0x00020000);

    unsigned voA[A_PW], voB[B_PW], amask[A_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int c = j * NW + wave;
        const int r = 8 * c + (lane >> 3);
        const int s = (lane & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        voA[j] = (c < A_CH && m < p.M) ? (unsigned)((long long)m * a_ld_bytes + 16 * s) : OOB;
        amask[j] = 0x1ffu;
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
            amask[j] = mk;
        }
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int c = j * NW + wave;
        const int r = 8 * c + (lane >> 3);
        const int s = (lane & 7) ^ ((r >> 1) & 7);
        const int n = n0 + r;
        voB[j] = (c < B_CH && n < p.N) ? (unsigned)((long long)n * p.ldb * EB + 16 * s) : OOB;
    }
    const int cgroups = (AMODE == A_CONV) ? p.Cin / BKE : 1;

    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;  // LDS byte address of the stages
    auto issue = [&](int kt, int stage) {
        uint4* sA = smem + stage * STAGE_SLOTS;
        uint4* sB = sA + BM * 8;
        unsigned soA, soB;
        int tap = 0;
        if (AMODE == A_CONV) {
            // channel-group-major K order: all 9 taps of one 32-channel slab back to back (the nine shifted re-reads
            // of an activation slab are adjacent in time: L1/L2 hits); the weights' K index is (tap, channel)
            tap = kt % 9;
            const int cg = kt / 9;
            soA = (unsigned)((((tap / 3) * p.W + (tap % 3)) * (long long)p.Cin * EB) + cg * 128);
            soB = (unsigned)((tap * cgroups + cg) * 128);
        } else {
            soA = soB = (unsigned)(kt * 128);
        }
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int c = j * NW + wave;
            if (A_CH % NW != 0 && c >= A_CH) break;
            unsigned vo = voA[j];
            if (AMODE == A_CONV) vo = ((amask[j] >> tap) & 1u) ? vo : OOB;
            dma16(rsA, sA + c * 64, vo, soA);
        }
#pragma unroll
        for (int j = 0; j < B_PW; ++j) {
            const int c = j * NW + wave;
            if (B_CH % NW != 0 && c >= B_CH) break;
            dma16(rsB, sB + c * 64, voB[j], soB);
        }
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int xs = (lane >> 1) & 7;
    const int a_row = wm * (32 * TM) + (lane & 31);
    const int b_row = wn * (32 * TN) + (lane & 31);

    auto compute = [&](int stage) {
        const uint4* sA = smem + stage * STAGE_SLOTS;
        const uint4* sB = sA + BM * 8;
        if constexpr (PL == 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[TM], b[TN];
                const int s = (2 * ks + khalf) ^ xs;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8, sA[(a_row + 32 * i) * 8 + s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8, sB[(b_row + 32 * j) * 8 + s]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[2][TM], b[2][TN];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int s = (4 * pl + 2 * ks + khalf) ^ xs;
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[pl][i] = __builtin_bit_cast(f16x8, sA[(a_row + 32 * i) * 8 + s]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[pl][j] = __builtin_bit_cast(f16x8, sB[(b_row + 32 * j) * 8 + s]);
                }
                // small terms first; consecutive MFMAs hit different accumulators
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop
    if constexpr (PP) {
        // Ping-pong: waves [0, NW/2) ("A") and [NW/2, NW) ("B") - one of each per SIMD - alternate between a READ period
        // (issue the next tile's LDS-DMA, pull the current tile's fragments LDS -> VGPRs) and an MFMA period (nothing but
        // the tile's MFMAs, at raised priority), B one period behind A: in every period one half feeds the matrix pipe
        // while the other half uses the LDS and the load path, instead of all waves doing the same thing at once.
        //   period 2t:   A reads tile t   | B multiplies tile t-1      (both issue tile t+1 -> stage (t+1)&1 first)
        //   period 2t+1: A multiplies t   | B reads tile t
        // one barrier between periods.  Stage (t+1)&1 held tile t-1: read by A in period 2t-2, by B in period 2t-1 with
        // its ds_reads retired (lgkmcnt(0)) before the barrier into period 2t.  Tile t+1 is first read in period 2t+2:
        // every wave retires its DMA share (vmcnt(0)) before the barrier into that period.
        static_assert(STAGES == 2 && NW % 2 == 0, "ping-pong schedule: two stages, an even number of waves");
        constexpr int NKS = PL == 2 ? 2 : 4, NPL = PL == 2 ? 2 : 1;
        uint4 fa[NKS][NPL][TM], fb[NKS][NPL][TN];
        auto rd = [&](int stage) {
            const uint4* sA = smem + stage * STAGE_SLOTS;
            const uint4* sB = sA + BM * 8;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const int s = ((PL == 2 ? 4 * pl : 0) + 2 * ks + khalf) ^ xs;
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[ks][pl][i] = sA[(a_row + 32 * i) * 8 + s];
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[ks][pl][j] = sB[(b_row + 32 * j) * 8 + s];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto mm = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if constexpr (PL == 1) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[ks][0][i]), __builtin_bit_cast(bf16x8, fb[ks][0][j]), acc[i][j], 0, 0, 0);
                } else {
                    // small terms first; consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ks][0][i]), __builtin_bit_cast(f16x8, fb[ks][1][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ks][1][i]), __builtin_bit_cast(f16x8, fb[ks][0][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ks][0][i]), __builtin_bit_cast(f16x8, fb[ks][0][j]), acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        auto bar = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (nk > 0) issue(kt_begin, 0);
        wait_vmcnt<0>();
        bar();  // tile 0 is in LDS
        if (wave < NW / 2) {
            for (int t = 0; t < nk; ++t) {
                if (t + 1 < nk) issue(kt_begin + t + 1, (t + 1) & 1);
                rd(t & 1);
                bar();
                mm();
                wait_vmcnt<0>();
                bar();
            }
            bar();
        } else {
            if (1 < nk) issue(kt_begin + 1, 1);
            bar();
            for (int t = 0; t < nk; ++t) {
                rd(t & 1);
                wait_vmcnt<0>();
                bar();
                if (t + 2 < nk) issue(kt_begin + t + 2, t & 1);
                mm();
                bar();
            }
        }
    } else if constexpr (SP != 0) {
        // Software pipeline (two LDS stages, P16 operands).  A K tile is two 16-deep MFMA steps; their operand fragments live
        // in two register sets, F0 (step 0) and F1 (step 1).  Per tile t, in program order (pinned by sched_barrier):
        //     wait: own LDS-DMA share of tile t landed (vmcnt 0), own F1 reads of tile t-1 returned (lgkmcnt 0);  s_barrier
        //     F0 <- LDS(tile t, step 0)
        //     MFMAs of tile t-1, step 1 (F1) - they cover the latency of the F0 reads - with the LDS-DMA pieces of tile t+1
        //         (into the stage tile t-1 occupied) in front of them (SP == 1) or spread between them (SP == 2)
        //     F1 <- LDS(tile t, step 1)
        //     MFMAs of tile t, step 0 (F0)   - they cover the latency of the F1 reads and the scalar address work of tile t+2
        // No MFMA waits for an LDS read issued right in front of it; what is exposed per tile is the barrier.  The products
        // reach every accumulator in the order of the plain loop (step 0, step 1 of tile t, then tile t+1): bit-identical
        // results.  Stage (t+1)&1 is free when its DMA is issued: the F0 and F1 reads of tile t-1 returned before this
        // tile's barrier.  The DMA goes through inline asm (dma16_asm): hipcc must not see it.
        static_assert(STAGES == 2 && PL == 2, "software pipeline: two stages, P16 operands");
        constexpr int NM = 3 * TM * TN;  // MFMAs per step
        const v4i_t rsA_s = raw_rsrc(a_base, (unsigned)(a_rows * a_ld_bytes)), rsB_s = raw_rsrc(Bp, (unsigned)((long long)p.N * p.ldb * EB));
        uint4 f0a[2][TM], f0b[2][TN], f1a[2][TM], f1b[2][TN];
        auto rd = [&](int stage, int ks, uint4 (&fa)[2][TM], uint4 (&fb)[2][TN]) {
            const uint4* sA = smem + stage * STAGE_SLOTS;
            const uint4* sB = sA + BM * 8;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int s = (4 * pl + 2 * ks + khalf) ^ xs;
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = sA[(a_row + 32 * i) * 8 + s];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[pl][j] = sB[(b_row + 32 * j) * 8 + s];
            }
        };
        // MFMA `idx` of a step: product idx / (TM TN) in the order (hi.lo, lo.hi, hi.hi) - small terms first -, accumulator idx % (TM TN)
        auto mm1 = [&](const uint4 (&fa)[2][TM], const uint4 (&fb)[2][TN], int idx) {
            const int pr = idx / (TM * TN), ij = idx % (TM * TN), i = ij / TN, j = ij % TN;
            const int pa = pr == 1 ? 1 : 0, pb = pr == 0 ? 1 : 0;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[pa][i]), __builtin_bit_cast(f16x8, fb[pb][j]), acc[i][j], 0, 0, 0);
        };
        // scalar state of the NEXT tile to issue: A / B byte offsets of its K tile and (3x3) its tap
        int n_tap = 0, n_dx = 0;
        unsigned n_soA = 0, n_soB = 0, n_rowA = 0, n_cgA = 0;
        const unsigned cin_b = (unsigned)p.Cin * EB, roww_b = (unsigned)p.W * cin_b;
        {
            const int kt = kt_begin;
            if (AMODE == A_CONV) {
                n_tap = kt % 9;
                const int cg = kt / 9;
                n_dx = n_tap % 3;
                n_rowA = (unsigned)(n_tap / 3) * roww_b;
                n_cgA = (unsigned)cg * 128u;
                n_soA = n_rowA + (unsigned)n_dx * cin_b + n_cgA;
                n_soB = (unsigned)(n_tap * cgroups + cg) * 128u;
            } else {
                n_soA = n_soB = (unsigned)kt * 128u;
            }
        }
        unsigned c_soA = 0, c_soB = 0;
        int c_tap = 0;
        auto latch = [&]() {  // the tile whose pieces are issued next
            c_soA = n_soA; c_soB = n_soB; c_tap = n_tap;
        };
        auto advance = [&]() {
            if (AMODE == A_CONV) {
                // (tap, channel group) -> next: taps innermost (gemm order: all 9 taps of a 32-channel slab back to back)
                ++n_tap; ++n_dx;
                n_soB += (unsigned)cgroups * 128u;
                if (n_dx == 3) { n_dx = 0; n_rowA += roww_b; }
                if (n_tap == 9) { n_tap = 0; n_rowA = 0; n_cgA += 128u; n_soB = n_soB - 9u * (unsigned)cgroups * 128u + 128u; }
                n_soA = n_rowA + (unsigned)n_dx * cin_b + n_cgA;
            } else {
                n_soA += 128u; n_soB += 128u;
            }
        };
        auto piece = [&](int stage, int q) {  // LDS-DMA instruction q of this wave's PER_TILE for the latched tile
            if (q < A_PW) {
                const int c = q * NW + wave;
                if (A_CH % NW != 0 && c >= A_CH) return;
                unsigned vo = voA[q];
                if (AMODE == A_CONV) vo = ((amask[q] >> c_tap) & 1u) ? vo : OOB;
                dma16_asm(rsA_s, smem_base + (unsigned)(stage * STAGE_SLOTS + c * 64) * 16u, vo, c_soA);
            } else {
                const int j = q - A_PW, c = j * NW + wave;
                if (B_CH % NW != 0 && c >= B_CH) return;
                dma16_asm(rsB_s, smem_base + (unsigned)(stage * STAGE_SLOTS + BM * 8 + c * 64) * 16u, voB[j], c_soB);
            }
        };
        auto tile = [&](int t, auto stage_c) {
            constexpr int stage = decltype(stage_c)::value;
            const bool has_prev = t > 0, has_next = t + 1 < nk;  // (wave-uniform)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);  // lgkmcnt(0) (vmcnt 0 too; the builtin so that hipcc's own count restarts here)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            rd(stage, 0, f0a, f0b);
            __builtin_amdgcn_sched_barrier(0);
            latch();
            if (has_next && ((SP & 3) == 1 || !has_prev)) {
#pragma unroll
                for (int q = 0; q < PER_TILE; ++q) piece(stage ^ 1, q);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((SP & 4) != 0) __builtin_amdgcn_s_setprio(1);  // (A/B: the MFMA phases above the other waves' loads)
            if (has_prev) {
                int q = 0;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    mm1(f1a, f1b, m);
                    if constexpr ((SP & 3) == 2) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < PER_TILE; ++e)
                            if (e == q && q * NM < (m + 1) * PER_TILE) {
                                if (has_next) piece(stage ^ 1, q);
                                ++q;
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            rd(stage, 1, f1a, f1b);
            __builtin_amdgcn_sched_barrier(0);
            advance();
#pragma unroll
            for (int m = 0; m < NM; ++m) mm1(f0a, f0b, m);
            if constexpr ((SP & 4) != 0) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        typedef std::integral_constant<int, 0> S0;
        typedef std::integral_constant<int, 1> S1;
        if (nk > 0) {
            latch();
#pragma unroll
            for (int q = 0; q < PER_TILE; ++q) piece(0, q);
            advance();
            int t = 0;
            for (; t + 1 < nk; t += 2) {
                tile(t, S0());
                tile(t + 1, S1());
            }
            if (t < nk) tile(t, S0());
#pragma unroll
            for (int m = 0; m < NM; ++m) mm1(f1a, f1b, m);
        }
    } else {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) issue(kt_begin + s, s);
        int stage = 0, istage = (STAGES - 1) % STAGES;
        for (int t = 0; t < nk; ++t) {
            // tile t has landed once at most `ahead` later tiles are still in flight
            const int ahead = min(STAGES - 2, nk - 1 - t);
            if (STAGES >= 4 && ahead == 2) wait_vmcnt<2 * PER_TILE>();
            else if (STAGES >= 3 && ahead >= 1) wait_vmcnt<PER_TILE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // every wave's share of tile t is in LDS; every wave is done with stage (t-1)
            asm volatile("" ::: "memory");
            if (t + STAGES - 1 < nk) issue(kt_begin + t + STAGES - 1, istage);
            compute(stage);
            stage = (stage + 1 == STAGES) ? 0 : stage + 1;
            istage = (istage + 1 == STAGES) ? 0 : istage + 1;
        }
    }

    // ---- epilogue (same contract as gemm_bf16.hip)
    const float unscale = 1.f / (scaleA * scaleB);
    const int row_base = m0 + wm * (32 * TM) + 4 * khalf;
    const int col_base = n0 + wn * (32 * TN) + (lane & 31);
    unsigned short* __restrict__ Cb = reinterpret_cast<unsigned short*>(p.C);  // c_fmt 2: plain bf16 output (batch = splits = 1)

    // BatchNorm partials from the FINAL values held in acc (accumulator layout)
    auto emit_stats = [&]() {
        // BatchNorm partials per 128-row slab of the tile (per TILE when BM is not a multiple of 128: the 96-row tiles,
        // whose partials cover 96 rows each - trid_gemm_p16_rows): column (mean, M2) over the slab's rows < M; with
        // stats_w == 4 also the column (min, max): the consumer derives max|BatchNorm(y)| - the fp16 scale of the
        // NEXT operand - from them before the apply pass runs (an affine map takes extremes to extremes).
        // Two levels, TWO barriers per tile (the one-level form - slab mean, then squared deviations from it - needed a
        // barrier pair per reduction: seven per tile, 3-13 % of the short-K layers): every wave takes (mean, M2, min, max)
        // of ITS 32 * TM rows in registers (its own mean: only means are subtracted), the waves of a slab meet in LDS once,
        // and one of them merges the wave partials in wave order with Chan's formula.
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(smem);  // [WM][BN]
        constexpr int SR = (BM % 128 == 0) ? 128 : BM;  // rows per slab = per partial
        constexpr int WPS = SR / (32 * TM);             // waves (along M) per slab
        constexpr int WR = 32 * TM;                     // rows per wave
        const int slab = wm / WPS;
        const int nw_i = min(max(p.M - (m0 + wm * WR), 0), WR);
        const float inv_nw = nw_i > 0 ? 1.f / (float)nw_i : 0.f;
        // (a tile that lies inside [0, M) - every tile of the benchmarked shapes - takes the loops without the per-element row
        // test: the same operations in the same order on the same values, half the VALU work of this epilogue)
        const bool whole = m0 + BM <= p.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = wn * (32 * TN) + 32 * j + (lane & 31);
            float s = 0.f, lo = INFINITY, hi = -INFINITY;
            if (whole) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        s += acc[i][j][r];
                        lo = fminf(lo, acc[i][j][r]);
                        hi = fmaxf(hi, acc[i][j][r]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (row < p.M) {
                            s += acc[i][j][r];
                            lo = fminf(lo, acc[i][j][r]);
                            hi = fmaxf(hi, acc[i][j][r]);
                        }
                    }
            }
            s += __shfl_xor(s, 32, 64);
            lo = fminf(lo, __shfl_xor(lo, 32, 64));
            hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
            const float mean = s * inv_nw;
            float q = 0.f;
            if (whole) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[i][j][r] - mean;
                        q += d * d;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                        const float d = acc[i][j][r] - mean;
                        if (row < p.M) q += d * d;
                    }
            }
            q += __shfl_xor(q, 32, 64);
            if (khalf == 0) red[wm * BN + cl] = make_float4(mean, q, lo, hi);
        }
        __syncthreads();
        if ((wm % WPS) == 0 && khalf == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = wn * (32 * TN) + 32 * j + (lane & 31);
                const int col = n0 + cl;
                float cnt = 0.f, mean = 0.f, m2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
                for (int w = 0; w < WPS; ++w) {
                    const int nb_i = min(max(p.M - (m0 + (slab * WPS + w) * WR), 0), WR);
                    if (nb_i > 0) {
                        const float4 v = red[(slab * WPS + w) * BN + cl];
                        const float nb = (float)nb_i, nt = cnt + nb, d = v.x - mean;
                        mean += d * (nb / nt);
                        m2 += v.y + d * d * (cnt * nb / nt);
                        cnt = nt;
                        lo = fminf(lo, v.z);
                        hi = fmaxf(hi, v.w);
                    }
                }
                if (col < p.N && cnt > 0.f) {
                    float* dst = p.stats + (((long long)mb * (BM / SR) + slab) * p.N + col) * p.stats_w;
                    dst[0] = mean;
                    dst[1] = m2;
                    if (p.stats_w == 4) {
                        dst[2] = lo;
                        dst[3] = hi;
                    }
                }
            }
        }
    };

    if constexpr (BM * BN * 4 <= p16_lds_bytes(BM, BN, WM * WN, STAGES)) {
        const bool pre = p.stats != nullptr;  // partials need the final values in the accumulator layout
        if (p.wide_epilogue && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.sC & 3) == 0 && (p.sSplit & 3) == 0 &&
            (p.res == nullptr || (p.ldres & 3) == 0) && !(pre && (p.accumulate || p.res != nullptr))) {
            // The tile goes through LDS so that a lane owns 4 consecutive columns and a wave stores whole rows of the tile
            // (16-byte pieces of fp32, 8-byte pieces of bf16) and reads `accumulate` / `res` the same way - the accumulator
            // layout itself has one column per lane: 4-byte (2-byte) stores, 128 (64) bytes per row and instruction.
            if (pre) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = col_base + 32 * j;
                    const float bv = (bias != nullptr && col < p.N) ? bias[col] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = p.alpha * unscale * acc[i][j][r] + bv;
                            v = p.relu ? fmaxf(v, 0.f) : v;
                            // a bf16 output: the partials are those of the tensor as stored
                            acc[i][j][r] = p.c_fmt == 2 ? __builtin_bit_cast(float, cvt_pk_bf16(v, 0.f) << 16) : v;
                        }
                }
                emit_stats();
            }
            __syncthreads();  // every wave is done with the operand stages - and with emit_stats' `red` (it does NOT end with a barrier)
            float* Ct = reinterpret_cast<float*>(smem);  // [BM][BN]
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = wm * (32 * TM) + 4 * khalf + i * 32 + (r & 3) + 8 * (r >> 2);
                        Ct[rl * BN + wn * (32 * TN) + 32 * j + (lane & 31)] = pre ? acc[i][j][r] : p.alpha * unscale * acc[i][j][r];
                    }
            __syncthreads();
            constexpr int C4 = BN / 4, RG = NW * 64 / C4;
            const int c4 = tid % C4, rg = tid / C4;
            const int col = n0 + 4 * c4;
            float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!pre && bias != nullptr && col < p.N) bv4 = *reinterpret_cast<const float4*>(bias + col);
            if (p.c_fmt == 1) {
                // eval-mode epilogue: out = act(colscale * y + bias (+ P16 residual)) written as a P16 tensor whose scale comes
                // from a bound every workgroup derives from the same device scalars (gemm_common.h EvalBound); the true
                // maximum of what was written is folded into *out_tmax for the next layer's bound
                const float bound = ev_bound;
                if (p.ev.out_bound != nullptr && blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) *p.ev.out_bound = bound;
                const float oscale = f16_scale_of(bound);
                const float rinv = ev_rinv;
                float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
                if (p.colscale != nullptr && col < p.N) cs = *reinterpret_cast<const float4*>(p.colscale + col);
                char* const outb = reinterpret_cast<char*>(p.C);
                const char* const resb = reinterpret_cast<const char*>(p.res16);
                unsigned am = 0;
                if (p.pool_w > 0) {
                    // ... followed by AvgPool2d(2) (a stride-2 block's conv2: m_resnet.py:59-61 under eval), written pooled: the
                    // tile's BM rows are BM / W whole image rows (W | BM, tiles start on even rows), i.e. BM / 4 pooled pixels that are
                    // consecutive in the pooled tensor; a lane averages the four activated values of its 4 columns
                    const int Wp = p.pool_w >> 1;
#pragma unroll
                    for (int ps = 0; ps < (BM / 4 + RG - 1) / RG; ++ps) {
                        const int pl = ps * RG + rg;  // pooled pixel of the tile
                        if (pl >= BM / 4 || col >= p.N) continue;
                        const int py = pl / Wp, px = pl - py * Wp;
                        const int r0 = 2 * py * p.pool_w + 2 * px;
                        if (m0 + r0 >= p.M) continue;
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int rl = r0 + (q >> 1) * p.pool_w + (q & 1);
                            float4 v = *reinterpret_cast<const float4*>(Ct + rl * BN + 4 * c4);
                            v = make_float4(fmaf(v.x, cs.x, bv4.x), fmaf(v.y, cs.y, bv4.y), fmaf(v.z, cs.z, bv4.z), fmaf(v.w, cs.w, bv4.w));
                            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                        }
                        a = make_float4(0.25f * a.x, 0.25f * a.y, 0.25f * a.z, 0.25f * a.w);
                        am = absmax4u(am, a);
                        const long long at = ((long long)(m0 >> 2) + pl) * p.N * 4 + (col >> 5) * 128 + (col & 31) * 2;
                        unsigned q0h, q0l, q1h, q1l;
                        f16_split2(a.x * oscale, a.y * oscale, q0h, q0l);
                        f16_split2(a.z * oscale, a.w * oscale, q1h, q1l);
                        p16_pair_store(outb + at, c4, q0h, q0l, q1h, q1l);
                    }
                } else
#pragma unroll
                for (int ps = 0; ps < BM / RG; ++ps) {
                    const int rl = ps * RG + rg, row = m0 + rl;
                    if (row >= p.M || col >= p.N) continue;
                    float4 v = *reinterpret_cast<const float4*>(Ct + rl * BN + 4 * c4);
                    v = make_float4(fmaf(v.x, cs.x, bv4.x), fmaf(v.y, cs.y, bv4.y), fmaf(v.z, cs.z, bv4.z), fmaf(v.w, cs.w, bv4.w));
                    // P16 row: 128 bytes per 32-column group = [hi x 32 | lo x 32]; this lane's 4 columns = 8 bytes of each plane
                    const long long at = (long long)row * p.N * 4 + (col >> 5) * 128 + (col & 31) * 2;
                    if (resb != nullptr) {
                        const uint2 h = *reinterpret_cast<const uint2*>(resb + at), l = *reinterpret_cast<const uint2*>(resb + at + 64);
                        const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
                        const f16x2 l0 = __builtin_bit_cast(f16x2, l.x), l1 = __builtin_bit_cast(f16x2, l.y);
                        v.x += ((float)h0.x + (float)l0.x) * rinv; v.y += ((float)h0.y + (float)l0.y) * rinv;
                        v.z += ((float)h1.x + (float)l1.x) * rinv; v.w += ((float)h1.y + (float)l1.y) * rinv;
                    }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    am = absmax4u(am, v);
                    unsigned q0h, q0l, q1h, q1l;
                    f16_split2(v.x * oscale, v.y * oscale, q0h, q0l);
                    f16_split2(v.z * oscale, v.w * oscale, q1h, q1l);
                    p16_pair_store(outb + at, c4, q0h, q0l, q1h, q1l);
                }
                // ONE atomic per workgroup, and only when it saw something above what the scalar held when it started (after the
                // first round of resident workgroups nearly none does).  Same-address device-scope atomics retire one by one,
                // ~15 ns each: one per wave (3072 tiles x 8) put 60 us of serialized atomics on a 180 us launch
                // (profiles/r05b_eval_calls.txt)
                am = wave_umax(am);
                __syncthreads();  // every wave is done reading Ct
                unsigned* redu = reinterpret_cast<unsigned*>(smem);
                if (lane == 0) redu[wave] = am;
                __syncthreads();
                if (tid == 0 && p.ev.out_tmax != nullptr) {
                    unsigned r = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) r = redu[w] > r ? redu[w] : r;
                    if (r > ev_seen) atomicMax(reinterpret_cast<unsigned*>(p.ev.out_tmax), r);
                }
                return;
            }
            // BatchNorm-backward sums of this tile's rows (gemm_common.h BnBwdFuse): this lane's 4 channels over its BM / RG rows.
            // Its eight arguments are read from the kernel-argument segment HERE, through a pointer the compiler cannot see
            // through: fetched at kernel entry like the others they stay live across the main loop, and the SGPRs this kernel
            // already spills into VGPR lanes cost it one more VGPR than its 128 (two waves per SIMD x 2 workgroups) hold
            unsigned long long kargs = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kargs));
            const BnBwdFuse bb = reinterpret_cast<const GemmParams*>(kargs)->bb;
            const bool bnb = bb.y != nullptr && col < p.N;
            float4 b_mu = make_float4(0.f, 0.f, 0.f, 0.f), b_is = b_mu, b_sc = b_mu, b_sh = b_mu;
            float4 b_s1 = b_mu, b_s2 = b_mu, b_mg = b_mu, b_mx = b_mu;
            if (bnb) {
                b_mu = *reinterpret_cast<const float4*>(bb.mean + col);
                b_is = *reinterpret_cast<const float4*>(bb.invstd + col);
                b_sc = *reinterpret_cast<const float4*>(bb.scale + col);
                b_sh = *reinterpret_cast<const float4*>(bb.shift + col);
            }
#pragma unroll
            for (int ps = 0; ps < BM / RG; ++ps) {
                const int rl = ps * RG + rg, row = m0 + rl;
                if (row >= p.M || col >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(Ct + rl * BN + 4 * c4);
                const long long at = (long long)row * p.ldc + col;
                float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bnb) yv = *reinterpret_cast<const float4*>(bb.y + (long long)row * p.N + col);  // (issued ahead of the stores)
                if (!pre) {
                    v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
                    if (p.accumulate) {
                        float4 u;
                        if (p.c_fmt == 2) {
                            const uint2 ub = *reinterpret_cast<const uint2*>(Cb + at);
                            u = make_float4(__builtin_bit_cast(float, ub.x << 16), __builtin_bit_cast(float, ub.x & 0xffff0000u),
                                            __builtin_bit_cast(float, ub.y << 16), __builtin_bit_cast(float, ub.y & 0xffff0000u));
                        } else {
                            u = *reinterpret_cast<const float4*>(C + at);
                        }
                        if (p.cmask != nullptr) {  // quad at / 4: four words per 64 quads (one per component), bit = quad % 64
                            const long long qi = at >> 2;
                            const ulonglong2* mq = reinterpret_cast<const ulonglong2*>(p.cmask + (qi >> 6) * 4);
                            const ulonglong2 m01 = mq[0], m23 = mq[1];
                            const int bit = (int)(qi & 63);
                            u.x = ((m01.x >> bit) & 1ull) ? u.x : 0.f; u.y = ((m01.y >> bit) & 1ull) ? u.y : 0.f;
                            u.z = ((m23.x >> bit) & 1ull) ? u.z : 0.f; u.w = ((m23.y >> bit) & 1ull) ? u.w : 0.f;
                        }
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    if (p.res != nullptr) {
                        const float4 rr = *reinterpret_cast<const float4*>(p.res + (long long)row * p.ldres + col);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                    }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                if (p.c_fmt == 2) *reinterpret_cast<uint2*>(Cb + at) = make_uint2(cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w));
                else *reinterpret_cast<float4*>(C + at) = v;
                if (bnb) {  // the arithmetic of bn_bwd_elem / bn_bwd_reduce_kernel (bn_pool.hip)
                    const float4 xh = make_float4((yv.x - b_mu.x) * b_is.x, (yv.y - b_mu.y) * b_is.y, (yv.z - b_mu.z) * b_is.z, (yv.w - b_mu.w) * b_is.w);
                    float4 g = v;
                    if (bb.relu == 1) {
                        g.x = fmaf(yv.x, b_sc.x, b_sh.x) > 0.f ? g.x : 0.f; g.y = fmaf(yv.y, b_sc.y, b_sh.y) > 0.f ? g.y : 0.f;
                        g.z = fmaf(yv.z, b_sc.z, b_sh.z) > 0.f ? g.z : 0.f; g.w = fmaf(yv.w, b_sc.w, b_sh.w) > 0.f ? g.w : 0.f;
                    } else if (bb.relu == 2) {  // the block output's ReLU bits (quad at / 4: four words per 64 quads, bit = quad % 64)
                        const long long qi = ((long long)row * p.N + col) >> 2;
                        const ulonglong2* mq = reinterpret_cast<const ulonglong2*>(bb.mask + (qi >> 6) * 4);
                        const ulonglong2 m01 = mq[0], m23 = mq[1];
                        const int bit = (int)(qi & 63);
                        g.x = ((m01.x >> bit) & 1ull) ? g.x : 0.f; g.y = ((m01.y >> bit) & 1ull) ? g.y : 0.f;
                        g.z = ((m23.x >> bit) & 1ull) ? g.z : 0.f; g.w = ((m23.y >> bit) & 1ull) ? g.w : 0.f;
                    }
                    b_s1.x += g.x; b_s1.y += g.y; b_s1.z += g.z; b_s1.w += g.w;
                    b_s2.x = fmaf(g.x, xh.x, b_s2.x); b_s2.y = fmaf(g.y, xh.y, b_s2.y); b_s2.z = fmaf(g.z, xh.z, b_s2.z); b_s2.w = fmaf(g.w, xh.w, b_s2.w);
                    b_mg.x = fmaxf(b_mg.x, fabsf(g.x)); b_mg.y = fmaxf(b_mg.y, fabsf(g.y)); b_mg.z = fmaxf(b_mg.z, fabsf(g.z)); b_mg.w = fmaxf(b_mg.w, fabsf(g.w));
                    b_mx.x = fmaxf(b_mx.x, fabsf(xh.x)); b_mx.y = fmaxf(b_mx.y, fabsf(xh.y)); b_mx.z = fmaxf(b_mx.z, fabsf(xh.z)); b_mx.w = fmaxf(b_mx.w, fabsf(xh.w));
                }
            }
            if (bb.y != nullptr) {
                // the RG row groups of a channel quad meet in LDS; the first C4 lanes fold them in row-group order (fixed order:
                // the partial is a function of the tile alone) and write the quad's entry of this tile's partial
                __syncthreads();  // every wave is done reading Ct
                float* redf = reinterpret_cast<float*>(smem);  // [RG][C4][16]
                float* mine = redf + (rg * C4 + c4) * 16;
                *reinterpret_cast<float4*>(mine) = b_s1;
                *reinterpret_cast<float4*>(mine + 4) = b_s2;
                *reinterpret_cast<float4*>(mine + 8) = b_mg;
                *reinterpret_cast<float4*>(mine + 12) = b_mx;
                __syncthreads();
                if (tid < C4 && n0 + 4 * tid < p.N) {
                    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mg = s1, mx = s1;
                    for (int r = 0; r < RG; ++r) {
                        const float* src = redf + (r * C4 + tid) * 16;
                        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                        const float4 c = *reinterpret_cast<const float4*>(src + 8), d = *reinterpret_cast<const float4*>(src + 12);
                        s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
                        s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
                        mg.x = fmaxf(mg.x, c.x); mg.y = fmaxf(mg.y, c.y); mg.z = fmaxf(mg.z, c.z); mg.w = fmaxf(mg.w, c.w);
                        mx.x = fmaxf(mx.x, d.x); mx.y = fmaxf(mx.y, d.y); mx.z = fmaxf(mx.z, d.z); mx.w = fmaxf(mx.w, d.w);
                    }
                    const int CQ = p.N >> 2, CW = CQ < 256 ? CQ : 256, S = CQ / CW;
                    const int q = (n0 >> 2) + tid;
                    const long long e = bn_bwd_partial_index(mb * S + q / CW, q % CW, p.mblocks * S, CQ);
                    *reinterpret_cast<float4*>(bb.ws + e) = s1;
                    *reinterpret_cast<float4*>(bb.ws + e + 4) = s2;
                    *reinterpret_cast<float4*>(bb.ws2 + e) = mg;
                    *reinterpret_cast<float4*>(bb.ws2 + e + 4) = mx;
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_base + 32 * j;
        const float bv = (bias != nullptr && col < p.N) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float oldv[16];
            if (p.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                    const long long at = (long long)row * p.ldc + col;
                    if (row < p.M && col < p.N) {
                        oldv[r] = p.c_fmt == 2 ? __builtin_bit_cast(float, (unsigned)Cb[at] << 16) : C[at];
                        if (p.cmask != nullptr && !((p.cmask[((at >> 2) >> 6) * 4 + (at & 3)] >> ((at >> 2) & 63)) & 1ull)) oldv[r] = 0.f;
                    } else {
                        oldv[r] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                float v = p.alpha * unscale * acc[i][j][r] + bv;
                if (p.accumulate) v += oldv[r];
                if (p.res != nullptr && row < p.M && col < p.N) v += p.res[(long long)row * p.ldres + col];
                if (p.relu) v = fmaxf(v, 0.f);
                if (row < p.M && col < p.N) {
                    const long long at = (long long)row * p.ldc + col;
                    if (p.c_fmt == 2) Cb[at] = (unsigned short)(cvt_pk_bf16(v, 0.f) & 0xffffu);  // round to nearest even
                    else C[at] = v;
                }
                acc[i][j][r] = p.c_fmt == 2 ? __builtin_bit_cast(float, cvt_pk_bf16(v, 0.f) << 16) : v;
            }
        }
    }

    if (p.stats != nullptr) emit_stats();
}

// ---------------------------------------------------------------------------------------------------- weight gradients
// dW[n][j] = sum over pixels m of dy[m][n] * X[m][j]: both operands are P16 tensors whose rows are PIXELS, i.e. the
// reduction index runs DOWN the rows while the MFMA wants 8 consecutive reduction indices per lane.  The tiles are
// staged exactly as they lie in HBM (LDS-DMA, one 128-byte (pixel, 32-channel group) piece per 8 lanes) and the
// transposition happens in the LDS READ: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4 x 16 tile
// of 16-bit elements whose rows are 4 consecutive pixels (row pitch 128 B here) - two such reads are one MFMA
// 32x32x16 operand fragment (8 pixels of one channel).  No VALU, no ds_write, no register transposes.
//
// LDS image of an operand stage: [32-channel group][32 pixels][128 B]; inside a piece the 16-byte slot s is stored at
// s ^ (4 * ((pixel >> 1) & 1)) (source-side permutation of the DMA), so the 32 lanes of a transpose read (4 pixels x
// 2 sixteen-channel halves x 4 eight-byte chunks) cover 256 distinct bytes of the 64 banks: conflict-free.
//
// B_NC: X = the layer input [pixels][C] (1x1 convolution);  B_CONV: X[m][(tap, c)] = x[m + tap offset][c] with
// zero padding (3x3 / stride 1 / pad 1): the DMA source row is shifted per 32-column group, invalid (pixel, tap)
// pairs read out of range = zeros.
typedef __fp16 h4v __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ f16x8 tr_frag(const char* lds_addr) {
    typedef __attribute__((address_space(3))) h4v* lp;
    const h4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_addr));
    const h4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_addr + 512));  // 4 pixels further (128 B each)
    f16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// PL = 2: P16 operands (pieces of 32 channels, 32-pixel K tiles); PL = 1: plain bf16 operands (pieces of 64 channels,
// 64-pixel K tiles, one bf16 MFMA per product).
// SP != 0: the software-pipelined main loop of gemm_p16_kernel (step-1 MFMAs of a tile after the next tile's barrier).
template <int BMODE, int BM, int STAGES, int PL, int SP = 0>
__global__ __launch_bounds__(512, 4) void gemm_p16_wgrad_kernel(GemmParams p) {
    constexpr int BN = 128, NW = 8, WN = 4, WM = 2;
    constexpr int GC = PL == 2 ? 32 : 64;        // channels per 128-byte piece
    constexpr int PBK = PL == 2 ? 32 : 64;       // pixels per K tile
    constexpr int EB = PL == 2 ? 4 : 2;          // bytes per element of a row
    constexpr int CPG = PBK / 8;                 // DMA chunks (8 pixels) per channel group
    constexpr int TM = BM / (32 * WM);           // 2 (BM = 128) or 1 (BM = 64)
    constexpr int A_CH = (BM / GC) * CPG, B_CH = (BN / GC) * CPG;  // 1-KB DMA chunks
    constexpr int A_PW = (A_CH + NW - 1) / NW, B_PW = B_CH / NW;
    constexpr int PER_TILE = A_PW + B_PW;
    constexpr int GBYTES = PBK * 128;            // one channel group's [pixels][128 B] block
    constexpr int A_BYTES = (BM / GC) * GBYTES, STAGE_BYTES = A_BYTES + (BN / GC) * GBYTES;
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    char* const sbase = reinterpret_cast<char*>(smem);

    const float scaleA = (PL == 2 && p.a_amax != nullptr) ? f16_scale_of(*p.a_amax) : 1.f;
    const float scaleB = (PL == 2 && p.b_amax != nullptr) ? f16_scale_of(*p.b_amax) : 1.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const uint32_t nwg = (uint32_t)p.mblocks * (uint32_t)p.nblocks;
    uint32_t lid;
    int sz;
    if (p.xcd_split) {
        // workgroup L runs on XCD L % 8: every XCD takes whole K splits (split s on XCD s % 8), so the tiles of a split -
        // which all read the same pixel range of dy and (shifted by their taps) of x - fetch it into ONE L2, once
        const uint32_t slot = blockIdx.x >> 3;
        sz = (int)((slot / nwg) * 8 + (blockIdx.x & 7));
        lid = slot % nwg;
        if (sz >= p.splits) return;
    } else {
        lid = xcd_remap(blockIdx.x, nwg);
        sz = blockIdx.z;
    }
    const int mb = lid / p.nblocks, nb = lid % p.nblocks;
    const int m0 = mb * BM, n0 = nb * BN;
    const int k_begin = sz * p.k_chunk;
    const int k_end = min(p.K, k_begin + p.k_chunk);
    const int nk = (k_end - k_begin + PBK - 1) / PBK;
    float* __restrict__ C = p.C + (long long)sz * p.sSplit;

    constexpr unsigned OOB = 0x80000000u;
    const char* A = reinterpret_cast<const char*>(p.A);
    const char* Bp = reinterpret_cast<const char*>(p.B);
    const long long a_ld = p.lda * EB, b_ld = (BMODE == B_CONV ? (long long)p.Cin : p.ldb) * EB;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((long long)p.K * a_ld), 0x00020000);
    const char* b_base = (BMODE == B_CONV) ? Bp - (long long)(p.W + 1) * b_ld : Bp;
    const long long b_rows = (BMODE == B_CONV) ? (long long)p.K + 2 * p.W + 2 : p.K;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (unsigned)(b_rows * b_ld), 0x00020000);

    // loader lanes: chunk c -> channel group c / CPG, pixels 8 * (c % CPG) + lane / 8, stored slot lane % 8
    unsigned voA[A_PW], voB[B_PW];
    int b_pp[B_PW], b_dy[B_PW], b_dx[B_PW];
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
        const int c = j * NW + wave;
        const int gi = c / CPG, pp = 8 * (c % CPG) + (lane >> 3);
        const int s = (lane & 7) ^ (((pp >> 1) & 1) << 2);
        voA[j] = (c < A_CH && m0 + GC * gi < p.M) ? (unsigned)((long long)pp * a_ld + (m0 + GC * gi) * EB + 16 * s) : OOB;
    }
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
        const int c = j * NW + wave;
        const int gi = c / CPG, pp = 8 * (c % CPG) + (lane >> 3);
        const int s = (lane & 7) ^ (((pp >> 1) & 1) << 2);
        const int col = n0 + GC * gi;
        b_pp[j] = pp;
        b_dy[j] = b_dx[j] = 0;
        if (BMODE == B_CONV) {
            const int tap = col / p.Cin, c0 = col - tap * p.Cin;
            b_dy[j] = tap / 3 - 1;
            b_dx[j] = tap % 3 - 1;
            // byte offset of (pixel pp + tap shift, channel group c0, slot s) from the shifted base
            voB[j] = col < p.N ? (unsigned)((long long)(pp + (b_dy[j] + 1) * p.W + (b_dx[j] + 1)) * b_ld + c0 * EB + 16 * s) : OOB;
        } else {
            voB[j] = col < p.N ? (unsigned)((long long)pp * b_ld + col * EB + 16 * s) : OOB;
        }
    }

    auto issue = [&](int kt, int stage) {
        char* sA = sbase + stage * STAGE_BYTES;
        char* sB = sA + A_BYTES;
        const int k0 = k_begin + kt * PBK;
        const unsigned soA = (unsigned)((long long)k0 * a_ld), soB = (unsigned)((long long)k0 * b_ld);
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int c = j * NW + wave;
            if (A_CH % NW != 0 && c >= A_CH) break;
            // rows beyond the split's range must not leak into it: the buffer only clips at K
            const unsigned vo = (k0 + 8 * (c % CPG) + (lane >> 3) < k_end) ? voA[j] : OOB;
            dma16(rsA, reinterpret_cast<uint4*>(sA + c * 1024), vo, soA);
        }
#pragma unroll
        for (int j = 0; j < B_PW; ++j) {
            const int c = j * NW + wave;
            unsigned vo = voB[j];
            const int m = k0 + b_pp[j];
            bool ok = m < k_end;
            if (BMODE == B_CONV) {
                const uint32_t q = fdiv((uint32_t)m, p.fdW);
                const int x = m - (int)q * p.W;
                const uint32_t b = fdiv(q, p.fdH);
                const int y = (int)q - (int)b * p.H;
                const int yy = y + b_dy[j], xx = x + b_dx[j];
                ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            }
            dma16(rsB, reinterpret_cast<uint4*>(sB + c * 1024), ok ? vo : OOB, soB);
        }
    };

    v16f acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // transpose-read lane constants: 16-channel half h, pixel row r of the 4 x 16 tile, 8-byte chunk c8, k half kh
    const int h = (lane >> 4) & 1, r4 = (lane & 15) >> 2, c8 = lane & 3, kh = lane >> 5;
    const int swz = (r4 >> 1) << 2;
    // byte offset inside a stage region of the fragment of 32-channel subtile `sub` (channels 32*sub .. +31 of the
    // tile), plane pl, k step ks: piece = (channel group, pixel 16 ks + 8 kh + r4), then the 16-byte slot
    auto frag_off = [&](int sub, int pl, int ks) {
        const int gi = (32 * sub) / GC;
        const int slot = PL == 2 ? (pl * 4 + h * 2 + (c8 >> 1)) : (((sub & 1) * 2 + h) * 2 + (c8 >> 1));
        return gi * GBYTES + (16 * ks + 8 * kh + r4) * 128 + ((slot ^ swz) << 4) + 8 * (c8 & 1);
    };

    auto compute = [&](int stage) {
        const char* sA = sbase + stage * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
        if constexpr (PL == 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[TM], b;
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8, tr_frag(sA + frag_off(wm * TM + i, 0, ks)));
                b = __builtin_bit_cast(bf16x8, tr_frag(sB + frag_off(wn, 0, ks)));
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[2][TM], b[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[pl][i] = tr_frag(sA + frag_off(wm * TM + i, pl, ks));
                    b[pl] = tr_frag(sB + frag_off(wn, pl, ks));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[1], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][i], b[0], acc[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][i], b[0], acc[i], 0, 0, 0);
            }
        }
    };

    // a wave whose 32 x 32 sub-tile lies outside [M] x [N] (32 output channels in a 64-row tile: the stem's conv2; the last
    // column tile of 9 * 32 = 288 columns) only loads and synchronises: its MFMAs would take half of its SIMD's matrix
    // pipe from the wave that has work
    const bool wave_live = (m0 + wm * (32 * TM) < p.M) && (n0 + wn * 32 < p.N);
    if constexpr (SP != 0) {
        static_assert(STAGES == 2 && PL == 2, "software pipeline: two stages, P16 operands");
        const v4i_t rsA_s = raw_rsrc(A, (unsigned)((long long)p.K * a_ld)), rsB_s = raw_rsrc(b_base, (unsigned)(b_rows * b_ld));
        const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
        // the LDS-DMA pieces of a K tile, out of the compiler's sight (dma16_asm; completion is counted at the top of a tile), in
        // two halves: prep(kt) works out every lane's source offset (row range of the split, 3x3: validity of the pixel's tap) -
        // placed under the MFMAs of the tile before -, fire(stage) is nothing but the DMA instructions
        unsigned nvA[A_PW], nvB[B_PW], n_soA = 0, n_soB = 0;
        auto prep = [&](int kt) {
            const int k0 = k_begin + kt * PBK;
            n_soA = (unsigned)((long long)k0 * a_ld);
            n_soB = (unsigned)((long long)k0 * b_ld);
#pragma unroll
            for (int j = 0; j < A_PW; ++j) {
                const int c = j * NW + wave;
                nvA[j] = (k0 + 8 * (c % CPG) + (lane >> 3) < k_end) ? voA[j] : OOB;
            }
#pragma unroll
            for (int j = 0; j < B_PW; ++j) {
                const int m = k0 + b_pp[j];
                bool ok = m < k_end;
                if (BMODE == B_CONV) {
                    const uint32_t q = fdiv((uint32_t)m, p.fdW);
                    const int x = m - (int)q * p.W;
                    const uint32_t b = fdiv(q, p.fdH);
                    const int y = (int)q - (int)b * p.H;
                    const int yy = y + b_dy[j], xx = x + b_dx[j];
                    ok = ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                }
                nvB[j] = ok ? voB[j] : OOB;
            }
        };
        auto fire = [&](int stage) {
            const unsigned sA = smem_base + (unsigned)(stage * STAGE_BYTES), sB = sA + A_BYTES;
#pragma unroll
            for (int j = 0; j < A_PW; ++j) {
                const int c = j * NW + wave;
                if (A_CH % NW != 0 && c >= A_CH) break;
                dma16_asm(rsA_s, sA + (unsigned)c * 1024u, nvA[j], n_soA);
            }
#pragma unroll
            for (int j = 0; j < B_PW; ++j) dma16_asm(rsB_s, sB + (unsigned)(j * NW + wave) * 1024u, nvB[j], n_soB);
        };
        f16x8 f0a[2][TM], f0b[2], f1a[2][TM], f1b[2];
        auto rd = [&](int stage, int ks, f16x8 (&fa)[2][TM], f16x8 (&fb)[2]) {
            const char* sA = sbase + stage * STAGE_BYTES;
            const char* sB = sA + A_BYTES;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = tr_frag(sA + frag_off(wm * TM + i, pl, ks));
                fb[pl] = tr_frag(sB + frag_off(wn, pl, ks));
            }
        };
        auto mm = [&](const f16x8 (&fa)[2][TM], const f16x8 (&fb)[2]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0], acc[i], 0, 0, 0);
        };
        auto tile = [&](int t, auto stage_c) {
            constexpr int stage = decltype(stage_c)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);  // lgkmcnt(0) (vmcnt 0 too; the builtin so that hipcc's own count restarts here)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (wave_live) rd(stage, 0, f0a, f0b);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nk) fire(stage ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if (wave_live) {
                if (t > 0) mm(f1a, f1b);
                __builtin_amdgcn_sched_barrier(0);
                rd(stage, 1, f1a, f1b);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) prep(t + 2);
            if (wave_live) mm(f0a, f0b);
            __builtin_amdgcn_sched_barrier(0);
        };
        typedef std::integral_constant<int, 0> S0;
        typedef std::integral_constant<int, 1> S1;
        if (nk > 0) {
            prep(0);
            fire(0);
            if (nk > 1) prep(1);
            int t = 0;
            for (; t + 1 < nk; t += 2) {
                tile(t, S0());
                tile(t + 1, S1());
            }
            if (t < nk) tile(t, S0());
            if (wave_live) mm(f1a, f1b);
        }
    } else {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) issue(s, s);
        int stage = 0, istage = (STAGES - 1) % STAGES;
        for (int t = 0; t < nk; ++t) {
            const int ahead = min(STAGES - 2, nk - 1 - t);
            if (STAGES >= 3 && ahead >= 1) wait_vmcnt<PER_TILE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + STAGES - 1 < nk) issue(t + STAGES - 1, istage);
            if (wave_live) compute(stage);
            stage = (stage + 1 == STAGES) ? 0 : stage + 1;
            istage = (istage + 1 == STAGES) ? 0 : istage + 1;
        }
    }

    const float f = p.alpha / (scaleA * scaleB);
    const int row_base = m0 + wm * (32 * TM) + 4 * kh;
    const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
            if (row < p.M && col < p.N) C[(long long)row * p.ldc + col] = f * acc[i][r];
        }
}

// ---------------------------------------------------------------------------------------------------- producers
// fmt 1: P16 (scaled fp16 planes); fmt 2: plain bf16 rows (round-to-nearest-even, no scale)
__device__ __forceinline__ void pack8_store(uint4* __restrict__ out, long long row, int K, int kg, const float (&v)[8], float scale, int fmt) {
    if (fmt == 2) {
        out[row * (K / 8) + kg] = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        return;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f16_split2(v[2 * q] * scale, v[2 * q + 1] * scale, h[q], l[q]);
    uint4* dst = out + row * (K / 4) + (kg >> 2) * 8 + (kg & 3);
    dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
    dst[4] = make_uint4(l[0], l[1], l[2], l[3]);
}

// fp32 [rows][K] (row pitch ldx) -> P16.  One thread = 8 consecutive k of one row.
__global__ __launch_bounds__(256) void p16_pack_kernel(const float* __restrict__ x, long long rows, int K, long long ldx,
                                                       const float* __restrict__ amax, uint4* __restrict__ out, int fmt) {
    const float scale = (fmt == 1 && amax != nullptr) ? f16_scale_of(*amax) : 1.f;
    const int K8 = K / 8;
    const long long total = rows * K8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long row = idx / K8;
        const int kg = (int)(idx - row * K8);
        const float4 u = *reinterpret_cast<const float4*>(x + row * ldx + 8 * kg);
        const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + 8 * kg + 4);
        const float x8[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        pack8_store(out, row, K, kg, x8, scale, fmt);
    }
}

// P16 -> fp32 (tests, consumers that need values): (hi + lo) / scale
__global__ __launch_bounds__(256) void p16_unpack_kernel(const uint4* __restrict__ in, long long rows, int K,
                                                         const float* __restrict__ amax, float* __restrict__ out) {
    const float inv = 1.f / (amax != nullptr ? f16_scale_of(*amax) : 1.f);
    const int K8 = K / 8;
    const long long total = rows * K8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long row = idx / K8;
        const int kg = (int)(idx - row * K8);
        const uint4* src = in + row * (K / 4) + (kg >> 2) * 8 + (kg & 3);
        const uint4 h = src[0], l = src[4];
        const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f16x2 a = __builtin_bit_cast(f16x2, hw[q]), b = __builtin_bit_cast(f16x2, lw[q]);
            v[2 * q] = ((float)a.x + (float)b.x) * inv;
            v[2 * q + 1] = ((float)a.y + (float)b.y) * inv;
        }
        float* dst = out + row * K + 8 * kg;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// w [N][T][C] fp32 -> P16 [C rows][K = T*N], k = t'*N + n with t' = flip ? T-1-t : t: the data-gradient operand of a
// convolution (3x3: 180-degree rotated, transposed filters; 1x1: the transposed matrix), packed in one pass.
__global__ __launch_bounds__(256) void p16_pack_wt_kernel(const float* __restrict__ w, int N, int T, int C, int flip,
                                                          const float* __restrict__ amax, uint4* __restrict__ out) {
    const float scale = amax != nullptr ? f16_scale_of(*amax) : 1.f;
    const int K = T * N, K8 = K / 8;
    const long long total = (long long)C * K8;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        // consecutive threads take consecutive c (the contiguous direction of w) for one 8-wide k group
        const int kg = (int)(idx / C);
        const int c = (int)(idx - (long long)kg * C);
        const int k0 = 8 * kg;
        const int tp = k0 / N, n0 = k0 - tp * N;
        const int t = flip ? T - 1 - tp : tp;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[((long long)(n0 + i) * T + t) * C + c] * scale;
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) f16_split2(v[2 * q], v[2 * q + 1], h[q], l[q]);
        uint4* dst = out + (long long)c * (K / 4) + (kg >> 2) * 8 + (kg & 3);
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[4] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

// Every conv filter of an encoder in ONE launch: table[t] = {src, dst, N, T, C, a} (device, 6 x int64 per tensor), amax[a]
// the tensor's largest magnitude.  transposed == 0: dst = P16 [N][K = T*C] (the forward operand, filters as stored);
// transposed != 0: dst = P16 [C][K = T*N] with the taps reversed for T > 1 (the data-gradient operand).
__global__ __launch_bounds__(256) void p16_pack_multi_kernel(const long long* __restrict__ table, const float* __restrict__ amax,
                                                             int transposed, int fmt) {
    const long long* e = table + 6 * (long long)blockIdx.y;
    const float* __restrict__ w = reinterpret_cast<const float*>(e[0]);
    uint4* __restrict__ out = reinterpret_cast<uint4*>(e[1]);
    const int N = (int)e[2], T = (int)e[3], C = (int)e[4];
    const float scale = fmt == 1 ? f16_scale_of(amax[e[5]]) : 1.f;
    if (!transposed) {
        const int K = T * C, K8 = K / 8;
        const long long total = (long long)N * K8;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
            const long long row = idx / K8;
            const int kg = (int)(idx - row * K8);
            const float4 u = *reinterpret_cast<const float4*>(w + row * K + 8 * kg);
            const float4 v = *reinterpret_cast<const float4*>(w + row * K + 8 * kg + 4);
            const float x8[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            pack8_store(out, row, K, kg, x8, scale, fmt);
        }
    } else {
        const int K = T * N, K8 = K / 8;
        const long long total = (long long)C * K8;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
            const int kg = (int)(idx / C);
            const int c = (int)(idx - (long long)kg * C);
            const int k0 = 8 * kg;
            const int tp = k0 / N, n0 = k0 - tp * N;
            const int t = T - 1 - tp;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = w[((long long)(n0 + i) * T + t) * C + c];
            pack8_store(out, c, K, kg, v, scale, fmt);
        }
    }
}

template <int AMODE, int BM, int BN, int WM, int WN, int STAGES, int PL = 2, bool PP = false, int SP = 0>
static int launch_p16(GemmParams& p, hipStream_t stream) {
    p.mblocks = (p.M + BM - 1) / BM;
    p.nblocks = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.mblocks * p.nblocks), 1, (unsigned)(p.batch * p.splits));
    // TRID_GEMM_LDS_PAD=<KB>: request at least that much LDS per workgroup - an occupancy knob for experiments (e.g. 96:
    // one workgroup per CU, which leaves registers and wave slots to the HBM-bound kernels of the other streams)
    static const size_t pad = getenv("TRID_GEMM_LDS_PAD") ? (size_t)atoi(getenv("TRID_GEMM_LDS_PAD")) * 1024 : 0;
    const size_t lds = std::max((size_t)p16_lds_bytes(BM, BN, WM * WN, STAGES), pad);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [lds] {
        if (lds > 48 * 1024)
            attr_err = hipFuncSetAttribute((const void*)gemm_p16_kernel<AMODE, BM, BN, WM, WN, STAGES, PL, PP, SP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_gemm_p16: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((gemm_p16_kernel<AMODE, BM, BN, WM, WN, STAGES, PL, PP, SP>), grid, dim3(WM * WN * 64), lds, stream, p);
    return check_launch("trid_gemm_p16");
}

// The 192-row, one-workgroup-per-CU tile (variant 15) as the library's own choice for a FORWARD convolution (BatchNorm-partials
// epilogue) whose 128x128 grid fills the 512 workgroup slots of the chip between half and fully - M = 24 576, N = 256: 384 tiles,
// every CU holds one or two workgroups and the launch lasts as long as the pairs do - while its 192x128 grid is a whole number of
// rounds of 256.  Alone the tile is +12 % (3x3, K = 2304) / +6 % (1x1, K = 1024) on exactly that shape (profiles/r06c_kloop.txt);
// INSIDE the step it is 0.09 ms SLOWER (three A/B rounds on one box, profiles/r06t_tile192_ab.txt: 39.92 vs 40.00 ms - a 12-wave
// workgroup holding a CU for itself shares it worse with the other lanes' kernels than two 8-wave workgroups do).
// OFF by default; TRID_P16_TILE192=1 switches the rule on (the partials then cover 192 rows each: trid_gemm_p16_rows).
static bool tile192_rule(int M, int N) {
    static const int env = getenv("TRID_P16_TILE192") ? atoi(getenv("TRID_P16_TILE192")) : 0;
    if (!env || M % 192 != 0 || N % 128 != 0) return false;
    const long long t128 = (long long)((M + 127) / 128) * (N / 128), t192 = (long long)(M / 192) * (N / 128);
    return t128 > 256 && t128 < 512 && t192 % 256 == 0;
}

template <int AMODE>
static int pick_p16(GemmParams& p, int variant, int planes, hipStream_t stream) {
    if (planes == 1) {  // bf16 operands: the residual blocks only (N >= 64)
        if (p.N <= 64) return launch_p16<AMODE, 128, 64, 2, 2, 3, 1>(p, stream);
        return launch_p16<AMODE, 128, 128, 2, 4, 2, 1>(p, stream);
    }
    if (p.N <= 32) return launch_p16<AMODE, 256, 32, 4, 1, 3>(p, stream);
    if (p.N <= 64) return launch_p16<AMODE, 128, 64, 2, 2, 3>(p, stream);
    // the library's choice: 8 waves of 64x32, two stages, software-pipelined main loop (profiles/r06c_kloop.txt: 0.43 of 833 on random /
    // 0.58 on zero operands over the layer2-4 3x3 shapes against 0.40 / 0.52 of variant 3, the barrier-per-tile loop it replaces;
    // TRID_P16_DEFAULT_VARIANT=3 brings that one back for A/B runs).  The eval epilogue and the BatchNorm-backward sums live in the
    // staged-through-LDS store path of the 128x128 8-wave tile: they take the default whatever was asked for.
    static const int def_env = getenv("TRID_P16_DEFAULT_VARIANT") ? atoi(getenv("TRID_P16_DEFAULT_VARIANT")) : 12;
    // (exactly the launches trid_gemm_p16_rows answers 192 for: P16 operands, N > 64, the statistics epilogue - which excludes the eval
    // epilogue, the BatchNorm-backward sums, batches and K splits -, no variant asked for)
    if (variant < 0 && p.stats != nullptr && tile192_rule(p.M, p.N)) variant = 15;
    const bool tile8 = variant == 3 || variant == 10 || variant == 12 || variant == 14;
    if (variant < 0 || ((p.c_fmt == 1 || p.bb.y != nullptr) && !tile8)) variant = def_env;
    switch (variant) {
        case 1: return launch_p16<AMODE, 128, 128, 2, 2, 3>(p, stream);   // 4 waves of 64x64, 3 stages (96 KB): 1 WG / CU
        case 2: return launch_p16<AMODE, 256, 128, 4, 2, 3>(p, stream);   // 8 waves of 64x64, 3 stages (144 KB)
        case 3: return launch_p16<AMODE, 128, 128, 2, 4, 2>(p, stream);   // 8 waves of 64x32, 2 stages (64 KB): 2 WG / CU
        case 4: return launch_p16<AMODE, 256, 128, 4, 2, 2>(p, stream);   // 8 waves of 64x64, 2 stages (96 KB)
        case 5: return launch_p16<AMODE, 128, 128, 2, 4, 3>(p, stream);   // 8 waves of 64x32, 3 stages (96 KB)
        case 8: return launch_p16<AMODE, 128, 64, 2, 2, 2>(p, stream);    // 4 waves of 64x32, 2 stages (48 KB): 3 WG / CU
        // 96-row tiles (6 waves of 32x64, 56 KB, 2 WG / CU): M = 24 576 x N = 256 becomes 512 workgroups = one full round of the
        // chip instead of 384 - measured 20-35 % SLOWER on every shape, balanced or not (profiles/r05a_tile96.txt: with 12
        // instead of 16 waves per CU the K loop hides less); kept selectable for that record, never chosen
        case 9: return launch_p16<AMODE, 96, 128, 3, 2, 2>(p, stream);
        case 6: return launch_p16<AMODE, 128, 128, 2, 4, 2, 2, true>(p, stream);  // variant 3 with the ping-pong schedule
        case 7: return launch_p16<AMODE, 256, 128, 4, 2, 2, 2, true>(p, stream);  // variant 4 with the ping-pong schedule
        case 10: return launch_p16<AMODE, 128, 128, 2, 4, 2, 2, false, 1>(p, stream);  // variant 3, software-pipelined main loop, DMA first
        case 11: return launch_p16<AMODE, 128, 128, 2, 2, 2, 2, false, 1>(p, stream);  // 4 waves of 64x64, software-pipelined, DMA first
        case 12: return launch_p16<AMODE, 128, 128, 2, 4, 2, 2, false, 2>(p, stream);  // variant 10 with the DMA pieces between the MFMAs
        case 13: return launch_p16<AMODE, 128, 128, 2, 2, 2, 2, false, 2>(p, stream);  // variant 11 with the DMA pieces between the MFMAs
        // 192-row tiles, 12 waves of 64x32, ONE workgroup per CU (80 KB of stages, 96 KB with the staged epilogue): every layer2-4
        // shape at B = 128 becomes a whole number of rounds of 256 workgroups (M = 24 576 x N = 512: 512 tiles instead of 768 on 512 slots)
        case 15: return launch_p16<AMODE, 192, 128, 3, 4, 2, 2, false, 2>(p, stream);
        case 16: return launch_p16<AMODE, 192, 128, 3, 4, 2, 2, false, 1>(p, stream);
        case 14: return launch_p16<AMODE, 128, 128, 2, 4, 2, 2, false, 6>(p, stream);  // variant 12 with raised priority from the first to the last MFMA of a tile
        default: return launch_p16<AMODE, 128, 128, 2, 2, 2>(p, stream);  // 4 waves of 64x64, 2 stages (64 KB): 2 WG / CU
    }
}

}  // namespace trid

using namespace trid;

extern "C" int trid_p16_pack_f32(const float* x, long long rows, int K, long long ldx, const float* amax, void* out, int fmt,
                                 void* stream) {
    TRID_REQUIRE(x && out && rows > 0 && K > 0 && K % 32 == 0 && ldx % 4 == 0 && aligned16(x) && aligned16(out) && (fmt == 1 || fmt == 2),
                 "trid_p16_pack_f32: needs K %% 32 == 0, 16-byte aligned rows and fmt 1 / 2 (K=%d)", K);
    hipLaunchKernelGGL(p16_pack_kernel, dim3(grid_for(rows * (K / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, rows, K,
                       ldx, amax, (uint4*)out, fmt);
    return check_launch("trid_p16_pack_f32");
}

extern "C" int trid_p16_unpack_f32(const void* in, long long rows, int K, const float* amax, float* out, void* stream) {
    TRID_REQUIRE(in && out && rows > 0 && K > 0 && K % 32 == 0 && aligned16(in) && aligned16(out), "trid_p16_unpack_f32: bad arguments");
    hipLaunchKernelGGL(p16_unpack_kernel, dim3(grid_for(rows * (K / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)in, rows, K, amax, out);
    return check_launch("trid_p16_unpack_f32");
}

extern "C" int trid_p16_pack_wt_f32(const float* w, int N, int T, int C, int flip, const float* amax, void* out, void* stream) {
    TRID_REQUIRE(w && out && N > 0 && T > 0 && C > 0 && N % 8 == 0 && (T * N) % 32 == 0 && aligned16(out),
                 "trid_p16_pack_wt_f32: needs N %% 8 == 0 and T*N %% 32 == 0 (N=%d T=%d)", N, T);
    hipLaunchKernelGGL(p16_pack_wt_kernel, dim3(grid_for((long long)C * (T * N / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       w, N, T, C, flip, amax, (uint4*)out);
    return check_launch("trid_p16_pack_wt_f32");
}

template <int BMODE, int BM, int PL, int SP = 0>
static int launch_p16_wgrad(GemmParams& p, hipStream_t stream) {
    constexpr int STAGES = 2;
    p.mblocks = (p.M + BM - 1) / BM;
    p.nblocks = (p.N + 127) / 128;
    static const int xcd_env = getenv("TRID_WGRAD_XCD") ? atoi(getenv("TRID_WGRAD_XCD")) : 1;  // (0: the plain tile-major grid, for A/B runs)
    p.xcd_split = (xcd_env && p.splits >= 8) ? 1 : 0;
    dim3 grid((unsigned)(p.mblocks * p.nblocks), 1, (unsigned)p.splits);
    if (p.xcd_split) grid = dim3((unsigned)(p.mblocks * p.nblocks) * 8u * (unsigned)((p.splits + 7) / 8), 1, 1);
    static const size_t pad = getenv("TRID_WGRAD_LDS_PAD") ? (size_t)atoi(getenv("TRID_WGRAD_LDS_PAD")) * 1024 : 0;  // (see launch_p16)
    const size_t lds = std::max((size_t)STAGES * ((BM / 32) + 4) * 4096, pad);  // (same bytes for both operand formats)
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [lds] {
        if (lds > 48 * 1024)
            attr_err = hipFuncSetAttribute((const void*)gemm_p16_wgrad_kernel<BMODE, BM, STAGES, PL, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_gemm_p16_wgrad: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((gemm_p16_wgrad_kernel<BMODE, BM, STAGES, PL, SP>), grid, dim3(512), lds, stream, p);
    return check_launch("trid_gemm_p16_wgrad");
}

extern "C" int trid_gemm_p16_wgrad(const trid_gemm_desc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TRID_REQUIRE(d != nullptr && d->A && d->B && d->C, "trid_gemm_p16_wgrad: null operand");
    TRID_REQUIRE(d->a_mode == A_MC && (d->b_mode == B_NC || d->b_mode == B_CONV), "trid_gemm_p16_wgrad: loader modes A_MC x B_NC / B_CONV only");
    const int planes = d->precision == 1 ? 1 : 2;  // precision 1: plain bf16 operands; else P16
    const int gc = planes == 1 ? 64 : 32;
    TRID_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->M % gc == 0 && d->N % gc == 0 && d->lda % gc == 0,
                 "trid_gemm_p16_wgrad: M, N and the row pitches must be multiples of %d (M=%d N=%d)", gc, d->M, d->N);
    TRID_REQUIRE(d->batch == 1 && d->splits >= 1 && !d->accumulate && !d->bias && !d->residual && !d->relu && !d->stats,
                 "trid_gemm_p16_wgrad: plain or split-K output only");
    TRID_REQUIRE(aligned16(d->A) && aligned16(d->B) && aligned16(d->C), "trid_gemm_p16_wgrad: operands must be 16-byte aligned");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = d->A; p.B = d->B; p.C = d->C;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.batch = 1; p.splits = d->splits;
    p.alpha = d->alpha;
    p.H = d->H; p.W = d->W; p.Cin = d->Cin;
    p.a_amax = d->a_amax; p.b_amax = d->b_amax;
    if (d->b_mode == B_CONV) {
        TRID_REQUIRE(d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % gc == 0 && d->N == 9 * d->Cin && d->K % (d->H * d->W) == 0,
                     "trid_gemm_p16_wgrad: B_CONV needs Cin %% %d == 0, N == 9*Cin, K a multiple of H*W", gc);
        p.fdW = make_fastdiv((uint32_t)d->W);
        p.fdH = make_fastdiv((uint32_t)d->H);
        TRID_REQUIRE((long long)(d->K + 2 * d->W + 2) * d->Cin * 4 < (1ll << 31), "trid_gemm_p16_wgrad: operands must stay below 2 GB");
    } else {
        TRID_REQUIRE(d->ldb % 32 == 0 && (long long)d->K * d->ldb * 4 < (1ll << 31), "trid_gemm_p16_wgrad: ldb %% 32 and B below 2 GB");
    }
    TRID_REQUIRE((long long)d->K * d->lda * 4 < (1ll << 31), "trid_gemm_p16_wgrad: operands must stay below 2 GB");
    const int pbk = planes == 1 ? 64 : 32;
    int kc = (d->K + d->splits - 1) / d->splits;
    kc = (kc + pbk - 1) / pbk * pbk;
    p.k_chunk = kc;
    p.sSplit = d->strideSplit;
    if (planes == 1) {
        if (d->b_mode == B_CONV) return d->M <= 64 ? launch_p16_wgrad<B_CONV, 64, 1>(p, stream) : launch_p16_wgrad<B_CONV, 128, 1>(p, stream);
        return d->M <= 64 ? launch_p16_wgrad<B_NC, 64, 1>(p, stream) : launch_p16_wgrad<B_NC, 128, 1>(p, stream);
    }
    static const int sp_env = getenv("TRID_WGRAD_SP") ? atoi(getenv("TRID_WGRAD_SP")) : 1;  // (the software-pipelined main loop; 0: the barrier-per-tile loop, A/B runs)
    if (sp_env && d->M > 64) return d->b_mode == B_CONV ? launch_p16_wgrad<B_CONV, 128, 2, 1>(p, stream) : launch_p16_wgrad<B_NC, 128, 2, 1>(p, stream);
    if (d->b_mode == B_CONV) return d->M <= 64 ? launch_p16_wgrad<B_CONV, 64, 2>(p, stream) : launch_p16_wgrad<B_CONV, 128, 2>(p, stream);
    return d->M <= 64 ? launch_p16_wgrad<B_NC, 64, 2>(p, stream) : launch_p16_wgrad<B_NC, 128, 2>(p, stream);
}

extern "C" int trid_p16_pack_multi_f32(const long long* table, const float* amax, int n_tensors, int transposed, int fmt, void* stream) {
    TRID_REQUIRE(table && (amax || fmt == 2) && n_tensors > 0 && (fmt == 1 || fmt == 2), "trid_p16_pack_multi_f32: bad arguments");
    hipLaunchKernelGGL(p16_pack_multi_kernel, dim3(64, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, table, amax, transposed, fmt);
    return check_launch("trid_p16_pack_multi_f32");
}

extern "C" int trid_gemm_p16_rows(int M, int N, int precision, int variant) {
    if (precision == 1 || N <= 64) return 128;
    if (variant < 0) return tile192_rule(M, N) ? 192 : 128;  // (what a statistics-epilogue launch of this shape will use)
    return variant == 9 ? 96 : ((variant == 15 || variant == 16) ? 192 : 128);
}

static int wide_env_ok() {
    static const int wide_env = getenv("TRID_GEMM_WIDE_EPILOGUE") ? atoi(getenv("TRID_GEMM_WIDE_EPILOGUE")) : 1;  // (0: A/B runs)
    return wide_env;
}

extern "C" int trid_gemm_p16_bnb_ok(int M, int N) {
    // (the shape half of trid_gemm_p16's precondition for bnb_y - the rest is about pointers - and the build-time / environment half)
    const int CQ = N / 4;
    return (M > 0 && N > 0 && (N == 64 || N % 128 == 0) && (256 % CQ == 0 || CQ % 256 == 0) && (long long)M * N * 4 < (1ll << 31) && wide_env_ok()) ? 1 : 0;
}

extern "C" int trid_gemm_p16(const trid_gemm_desc* d, int variant, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TRID_REQUIRE(d != nullptr && d->A && d->B && d->C, "trid_gemm_p16: null operand");
    const int planes = d->precision == 1 ? 1 : 2;  // precision 1: plain bf16 operands (64-wide K tiles); else P16
    const int bke = planes == 1 ? 64 : 32;
    TRID_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->K % bke == 0, "trid_gemm_p16: K must be a positive multiple of %d (K=%d)", bke, d->K);
    TRID_REQUIRE((d->a_mode == A_KC || d->a_mode == A_CONV) && d->b_mode == B_KC, "trid_gemm_p16: loader modes A_KC / A_CONV x B_KC only");
    TRID_REQUIRE(aligned16(d->A) && aligned16(d->B) && aligned16(d->C), "trid_gemm_p16: operands must be 16-byte aligned");
    TRID_REQUIRE(d->batch >= 1 && d->splits >= 1, "trid_gemm_p16: batch/splits must be >= 1");
    TRID_REQUIRE(d->lda % 32 == 0 && d->ldb % 32 == 0, "trid_gemm_p16: row pitches must be multiples of 32 elements");
    TRID_REQUIRE(planes == 2 || d->N >= 64, "trid_gemm_p16: bf16 operands need N >= 64");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = d->A; p.B = d->B; p.C = d->C;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
    p.sA = d->strideA; p.sB = d->strideB; p.sC = d->strideC;
    p.batch = d->batch; p.splits = d->splits;
    p.alpha = d->alpha; p.accumulate = d->accumulate;
    p.bias = d->bias; p.sBias = d->strideBias; p.stats = d->stats;
    p.res = d->residual; p.ldres = d->ldres; p.relu = d->relu;
    p.H = d->H; p.W = d->W; p.Cin = d->Cin;
    p.a_amax = d->a_amax; p.b_amax = d->b_amax;
    p.stats_w = d->stats_minmax ? 4 : 2;
    p.c_fmt = d->c_format;
    p.cmask = reinterpret_cast<const unsigned long long*>(d->c_mask);
    p.colscale = d->col_scale; p.res16 = d->res_p16; p.res16_amax = d->res_amax;
    p.ev.coef = d->eval_coef; p.ev.tin = d->eval_tin; p.ev.tres = d->eval_tres;
    p.ev.out_bound = d->out_bound; p.ev.out_tmax = d->out_tmax;
    p.pool_w = d->eval_pool_w;
    p.bb.y = d->bnb_y; p.bb.mean = d->bnb_mean; p.bb.invstd = d->bnb_invstd; p.bb.scale = d->bnb_scale; p.bb.shift = d->bnb_shift;
    p.bb.ws = d->bnb_ws; p.bb.ws2 = d->bnb_ws2; p.bb.relu = d->bnb_relu;
    p.bb.mask = reinterpret_cast<const unsigned long long*>(d->bnb_mask);
    if (d->bnb_y != nullptr) {
        const int CQ = d->N / 4;
        TRID_REQUIRE(planes == 2 && d->c_format == 0 && d->batch == 1 && d->splits == 1 && d->ldc == d->N && (d->N == 64 || d->N % 128 == 0) && !d->stats &&
                     (256 % CQ == 0 || CQ % 256 == 0) && wide_env_ok(),
                     "trid_gemm_p16: the BatchNorm-backward sums (bnb_y) need P16 operands, fp32 C with ldc == N, N == 64 or N %% 128 == 0, batch == splits == 1, no stats (N=%d)", d->N);
        TRID_REQUIRE(d->bnb_relu >= 0 && d->bnb_relu <= 2 && (d->bnb_relu != 2 || (d->bnb_mask != nullptr && aligned16(d->bnb_mask))),
                     "trid_gemm_p16: bnb_relu must be 0, 1 or 2 (2: with the 16-byte aligned ReLU bit mask bnb_mask)");
        TRID_REQUIRE(d->bnb_mean && d->bnb_invstd && d->bnb_scale && d->bnb_shift && d->bnb_ws && d->bnb_ws2 && aligned16(d->bnb_y) && aligned16(d->bnb_mean) &&
                     aligned16(d->bnb_invstd) && aligned16(d->bnb_scale) && aligned16(d->bnb_shift) && aligned16(d->bnb_ws) && aligned16(d->bnb_ws2),
                     "trid_gemm_p16: the BatchNorm-backward sums need all of bnb_mean / invstd / scale / shift / ws / ws2, 16-byte aligned");
    }
    TRID_REQUIRE(p.cmask == nullptr || (d->accumulate && d->batch == 1 && d->splits == 1 && d->ldc == d->N && d->N % 4 == 0 && !d->stats),
                 "trid_gemm_p16: c_mask needs accumulate, batch == splits == 1, ldc == N, N %% 4 == 0");
    p.wide_epilogue = wide_env_ok();
    TRID_REQUIRE(p.c_fmt == 0 || ((p.c_fmt == 2 || p.c_fmt == 1) && d->batch == 1 && d->splits == 1),
                 "trid_gemm_p16: c_format must be 0, or 1 / 2 with batch == splits == 1");
    if (p.c_fmt == 1) {
        TRID_REQUIRE(planes == 2 && !d->accumulate && !d->stats && !d->residual && d->ldc == d->N && d->N % 32 == 0 && (long long)d->M * d->N * 4 < (1ll << 31),
                     "trid_gemm_p16: the eval epilogue (c_format 1) needs P16 operands, ldc == N, N %% 32 == 0, no accumulate / stats / fp32 residual (N=%d)", d->N);
        TRID_REQUIRE(d->eval_coef && d->eval_tin && (!d->res_p16 || (d->res_amax && d->eval_tres && aligned16(d->res_p16))) && (!d->col_scale || aligned16(d->col_scale)) &&
                     (!d->bias || aligned16(d->bias)), "trid_gemm_p16: the eval epilogue needs eval_coef / eval_tin (and res_amax / eval_tres with a residual), 16-byte aligned vectors");
        p.wide_epilogue = 1;  // (the only form of this epilogue)
        if (d->eval_pool_w) {
            TRID_REQUIRE(d->a_mode == A_CONV && d->eval_pool_w == d->W && d->W % 2 == 0 && d->H % 2 == 0 && 128 % d->W == 0 && (128 / d->W) % 2 == 0 &&
                         (d->H * d->W) % 128 == 0 && d->N > 64 && !d->res_p16,
                         "trid_gemm_p16: the pooled eval epilogue needs a 3x3 convolution with W | 128, an even number of image rows per 128-row tile, H * W %% 128 == 0, N > 64, no residual (H=%d W=%d N=%d)", d->H, d->W, d->N);
        }
    } else {
        TRID_REQUIRE(!d->col_scale && !d->res_p16, "trid_gemm_p16: col_scale / res_p16 belong to the eval epilogue (c_format 1)");
    }
    if (d->a_mode == A_CONV) {
        TRID_REQUIRE(d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % bke == 0 && d->K == 9 * d->Cin && d->M % (d->H * d->W) == 0 && d->splits == 1,
                     "trid_gemm_p16: A_CONV needs Cin %% %d == 0, K == 9*Cin, M a multiple of H*W, splits == 1", bke);
        p.fdW = make_fastdiv((uint32_t)d->W);
        p.fdH = make_fastdiv((uint32_t)d->H);
    }
    TRID_REQUIRE(!(d->stats && (d->splits != 1 || d->batch != 1)), "trid_gemm_p16: stats epilogue needs splits==1, batch==1");
    TRID_REQUIRE(!(d->splits > 1 && (d->accumulate || d->bias || d->residual || d->relu)), "trid_gemm_p16: split-K writes raw slabs");
    const long long a_bytes = (d->a_mode == A_CONV ? (long long)(d->M + 2 * d->W + 2) * d->Cin : (long long)d->M * d->lda) * 4;
    TRID_REQUIRE(a_bytes < (1ll << 31) && (long long)d->N * d->ldb * 4 < (1ll << 31), "trid_gemm_p16: operands must stay below 2 GB (31-bit buffer offsets)");
    int kc = (d->K + d->splits - 1) / d->splits;
    kc = (kc + bke - 1) / bke * bke;
    p.k_chunk = kc;
    p.sSplit = d->strideSplit;
    if (d->a_mode == A_CONV) return pick_p16<A_CONV>(p, variant, planes, stream);
    return pick_p16<A_KC>(p, variant, planes, stream);
}
