// Short-K 1x1 convolutions (K = 64 / 128 / 256 input channels) as a STREAMING kernel.
//
// C[M][N] = A[M][K] . B[N][K]^T with both operands P16 (gemm_p16.hip).  In the residual blocks these are the expand
// convolutions conv3 / downsample of layer1-3 (m_resnet.py:26,41-47: K = planes, N = 4 planes) and the data gradients
// of conv1 (K = planes): 4 K bytes read and 4 N bytes written per row against 6 K N flops - HBM-bound by construction
// (layer1 conv3: 100 MB in, 403 MB out, 12.9 GFLOP).  A tile kernel whose K loop is two tiles long spends its time in
// prologue, BatchNorm-partial barriers and an epilogue that nothing overlaps; here
//   * the FILTER never touches LDS: a wave owns 32 output columns and keeps their whole [32][K] panel as MFMA B
//     fragments in K/2 VGPRs for the lifetime of the (persistent) workgroup;
//   * the workgroup walks down the rows: each step brings the next RB x K activation tile in by LDS-DMA (two stages, one
//     barrier per step, 16-byte units XOR-swizzled by the row: conflict-free fragment reads) while the MFMAs of the
//     current tile run and its output drains;
//   * with 8 column waves (N >= 256) a wave sees ALL rows of a step for its columns: the BatchNorm (mean, M2, min, max)
//     partials are taken in registers and stored by the wave itself - no barrier, no LDS;
//   * stores go out as whole 128-byte segments of C rows straight from the accumulator layout.
// Arithmetic and product order are those of gemm_p16_kernel<A_KC> (hi*lo + lo*hi + hi*hi per 16-deep k step, k ascending):
// results are bit-identical.

#include <algorithm>
#include <mutex>

#include "split_common.h"

#ifndef TRID_TOPK_PF
#define TRID_TOPK_PF 1  // fragment prefetch distance of the retrieval filter instantiation (k steps); 2 measured the same (16.4 ms) at 256 VGPRs
#endif

namespace trid {

namespace {

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// (nt: the activation tile is read once by one workgroup and the output written once - non-temporal for tensors far beyond
// the Infinity Cache, split_common.h STREAM_NT_MIN_BYTES)
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& rs, void* lds_base, unsigned voffset, bool nt) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    if (nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_base, 16, voffset, 0, 0, 2);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ void store_c(unsigned v, const __amdgpu_buffer_rsrc_t& rs, unsigned off, bool nt) {
    if (nt) __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0);
}
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
// (an LDS store the compiler does not see: beside an LDS-DMA in flight it would be ordered behind s_waitcnt vmcnt(0))
__device__ __forceinline__ void lds_store4(const void* p, float4 v) {
    const v4f r = {v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(uintptr_t)p), "v"(r) : "memory");
}

struct StreamParams {
    const char* A;   // P16 [M][K]
    const char* B;   // P16 [N][K]
    float* C;        // fp32 [M][ldc]
    float* stats;    // [ceil(M / RB)][N][4] = (mean, M2, min, max) per step tile, or null
    const float* a_amax;
    const float* b_amax;
    int M, N;
    long long ldc;
    int accumulate;
    int panels;      // column panels of CW * 32 columns
    int workers;     // persistent workgroups per panel
    int tiles;       // ceil(M / RB)
    int nt_a, nt_c;  // non-temporal activation loads / output stores
    const unsigned* cmask;  // accumulate only, ldc == N, N % 256 == 0: bit mask over C (bn_apply's relu_mask layout) gating the OLD values
    // FUSE (trid_conv1x1_bn_res_p16): the BatchNorm + identity residual + ReLU of bn_apply_kernel<1, 1> in the epilogue
    const float* bn_scale;   // [N]
    const float* bn_shift;   // [N]
    const char* res;         // P16 [M][N]: the identity branch
    const float* res_amax;
    char* out;               // P16 [M][N]
    const float* oa;         // bound of max|bn(y)| and of the residual: the output's scale comes from their sum ...
    const float* ob;
    float* osum;             // ... which is published here
    unsigned char* mask;     // relu_mask words of the output (1 bit per element), or null
    int relu;
    int nt_o;
    EvalBound ev;            // eval mode (trid_conv1x1_eval_p16; ev.coef != null): the output's scale from the analytic bound, res optional
    // FUSE == 4 (retrieval, stream_topk_filter): nothing is stored but the elements that reach their COLUMN's threshold
    // (columns = queries, rows = gallery rows): appended to the column's candidate list through an atomic counter
    GemmFilter filt;
    int n_real;              // columns that exist (N is padded to a multiple of 32)
    int flags;               // FUSE == 4: 1 = DMA pieces spread over the k loop, 2 = the upper waves scan one step late
};

__device__ __forceinline__ void lds_store1(const void* p, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(uintptr_t)p), "v"(v) : "memory");
}
// (inline asm for the same reason: beside an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) before a visible LDS access)
__device__ __forceinline__ unsigned lds_add_rtn(unsigned addr, unsigned v) {
    unsigned r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ unsigned lds_load1(unsigned addr) {
    unsigned r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ void lds_store1u(unsigned addr, unsigned v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void filter_append(const GemmFilter& f, int col, float v, int row) {
    const int slot = atomicAdd(f.cnt + col, 1);
    if (slot < f.cap) reinterpret_cast<float2*>(f.cand)[(long long)col * f.cap + slot] = make_float2(v, __int_as_float(row));
    else *f.overflow = 1;
}

}  // namespace

// K: reduction length; CW: column waves (each 32 columns); TM: 32-row blocks per wave and step.  8 waves = CW x RW.
// FUSE (CW = 8, no accumulate): 0 = store C; 1 = C is not stored, the tile goes through BatchNorm + residual + ReLU and is
// written as a P16 tensor (+ ReLU bit mask); 2 = both (the raw conv output is kept for the backward pass).
template <int K, int CW, int TM, bool ACC, int FUSE = 0>
__global__ __launch_bounds__(512, 2) void gemm_p16_stream_kernel(StreamParams p) {
    static_assert(FUSE == 0 || (CW == 8 && !ACC), "the fused epilogue is built for the 8-column-wave, non-accumulating form");
    // FUSE == 4: the top-k admission filter of the retrieval match (evaluation.py:117-120, 17-19): the QUERIES are the column
    // panel a wave keeps in registers (32 queries x K = 256 in 128 VGPRs), the gallery streams through the LDS ring once per
    // 256 queries, and an element leaves the chip only when it reaches its query's admission threshold (one register)
    // FUSE == 3: the statistics-only pass (nothing is stored but the partials).  Without the output stream a step is over in
    // a fraction of the HBM latency, so one tile in flight per workgroup starves it (66 us for 100 MB): THREE stages, two
    // tiles in flight.  (With C stores in between the 6-bit in-order vmcnt could not tell the older DMA from the newer one.)
    constexpr int NS = FUSE == 3 ? 3 : 2;
    constexpr int NW = 8, RW = NW / CW;
    constexpr int RB = RW * TM * 32;          // rows per step
    constexpr int ROWB = K * 4;               // bytes of one A row
    constexpr int UPR = ROWB / 16;            // 16-byte units per row (16 / 32 / 64)
    constexpr int RPC = 1024 / ROWB;          // rows per 1-KB DMA chunk (4 / 2 / 1)
    constexpr int STAGE = RB * ROWB;          // bytes of one stage
    constexpr int NCH = STAGE / 1024;         // DMA chunks per stage
    constexpr int DPW = NCH / NW;             // DMA instructions per wave and step
    constexpr int KG = K / 32;
    static_assert(NCH % NW == 0, "stage must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    char* const ring = reinterpret_cast<char*>(smem);
    float4* const sstat = reinterpret_cast<float4*>(ring + NS * STAGE);  // RW > 1: [2][NW][32]
    constexpr int TP = 36;                                                 // FUSE: row pitch (floats) of a wave's transposition tile
    float* const tbuf = reinterpret_cast<float*>(sstat + 2 * NW * 32);     // FUSE: [NW][32][TP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave % CW, rw = wave / CW;
    const int khalf = lane >> 5;

    // workgroup -> (panel, worker): the panels of one worker index share an XCD (block b runs on XCD b % 8) and walk the same
    // row tiles, so an activation tile is fetched from HBM once and found in that L2 by the other panels
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int panel = idx % p.panels;
    const int worker = xcd + 8 * (idx / p.panels);
    if (worker >= p.workers) return;
    const int n0 = panel * (CW * 32) + cw * 32;
    const bool col_live = n0 < p.N;  // (N % 32 == 0: a wave's 32 columns exist or do not)

    const float unscale = 1.f / (f16_scale_of(*p.a_amax) * f16_scale_of(*p.b_amax));

    // ---- this wave's filter panel: [k group][k step][plane] fragments, K/2 VGPRs
    f16x8 bf[KG][2][2];
    {
        const char* br = p.B + (size_t)((col_live ? n0 : 0) + (lane & 31)) * ROWB;
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    bf[g][ks][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(br + g * 128 + (4 * pl + 2 * ks + khalf) * 16));
    }

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)((size_t)p.M * ROWB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (unsigned)((size_t)p.M * p.ldc * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.stats != nullptr ? p.stats : p.C), 0, (unsigned)((size_t)p.tiles * p.N * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)(p.cmask != nullptr ? (const void*)p.cmask : (const void*)p.C), 0, (unsigned)((size_t)p.M * p.N / 8), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // FUSE: this lane's 8 channels (octet oc of the wave's 32 columns; the same in both halves of a tile) and their coefficients
    constexpr bool FEPI = FUSE == 1 || FUSE == 2;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(FEPI ? (const void*)p.res : (const void*)p.C), 0, (unsigned)((size_t)p.M * p.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(FEPI ? (void*)p.out : (void*)p.C), 0, (unsigned)((size_t)p.M * p.N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)((FEPI && p.mask != nullptr) ? (void*)p.mask : (void*)p.C), 0, (unsigned)((size_t)p.M * p.N / 8), 0x00020000);
    float fsc[8], fsh[8];
    float oscale = 1.f, rinv = 1.f;
    if constexpr (FUSE == 1 || FUSE == 2) {
        const int c0 = (col_live ? n0 : 0) + 8 * (lane & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            fsc[j] = p.bn_scale[c0 + j];
            fsh[j] = p.bn_shift[c0 + j];
        }
        float bound;
        if (p.ev.coef != nullptr) {
            bound = eval_out_bound(p.ev);
            if (p.ev.out_bound != nullptr && blockIdx.x == 0 && tid == 0) *p.ev.out_bound = bound;
        } else {
            bound = (p.oa != nullptr ? *p.oa : 0.f) + (p.ob != nullptr ? *p.ob : 0.f);
            if (p.osum != nullptr && blockIdx.x == 0 && tid == 0) *p.osum = bound;
        }
        oscale = f16_scale_of(bound);
        rinv = p.res != nullptr ? 1.f / f16_scale_of(*p.res_amax) : 1.f;
    }
    unsigned tmax_bits = 0;  // eval mode: true max|out| of everything this lane wrote
    float f_thr = INFINITY;  // FUSE == 4: this lane's column (query) threshold
    if constexpr (FUSE == 4) {
        const int col = n0 + (lane & 31);
        if (col_live && col < p.n_real) f_thr = p.filt.thr[(long long)col * p.filt.thr_stride];
    }
    // FUSE == 4: this wave's staging list of candidates, [FCAP] x (value, gallery row, query), and its counter
    constexpr int FCAP = 128;
    // (LDS byte addresses held in ONE vector register each: handed to the inline asm as wave-uniform values the compiler makes
    // a copy per use and hoists all of them out of the loop - 33 spilled registers)
    unsigned f_wbuf = (unsigned)(uintptr_t)(tbuf + wave * (3 * FCAP));
    unsigned f_wcnt = (unsigned)(uintptr_t)(reinterpret_cast<unsigned*>(sstat) + wave);
    if constexpr (FUSE == 4) {
        asm volatile("" : "+v"(f_wbuf), "+v"(f_wcnt));
        if (lane == 0) lds_store1u(f_wcnt, 0u);
    }
    auto flush_staged = [&](unsigned staged) {  // wave-uniform: the whole wave empties its list, one candidate per lane and round
        const unsigned n = staged < (unsigned)FCAP ? staged : (unsigned)FCAP;
        for (unsigned e = lane; e < n; e += 64) {
            const float v = __uint_as_float(lds_load1(f_wbuf + 12u * e));
            const int row = (int)lds_load1(f_wbuf + 12u * e + 4u), col = (int)lds_load1(f_wbuf + 12u * e + 8u);
            filter_append(p.filt, col, v, row);
        }
        if (lane == 0) lds_store1u(f_wcnt, 0u);  // (the wave's LDS operations execute in order: behind the reads above)
    };

    // loader: chunk c (1 KB = RPC rows) of a stage; lane -> (row, stored unit j); source unit = j ^ (row & 15).  The per-chunk
    // offsets are re-derived from the lane index in every step (`zero` is opaque to the compiler): kept across the loop
    // they would cost DPW x 2 VGPRs of a budget that the filter panel (K/2) and the accumulators already fill
    auto issue = [&](int tile, int stage, int zero) {
        const size_t base = (size_t)tile * RB * ROWB;
        char* dst = ring + stage * STAGE;
        const int ln = lane + zero;
        const int lr = ln / UPR, j = ln % UPR;
#pragma unroll
        for (int d = 0; d < DPW; ++d) {
            const int r = (d * NW + wave) * RPC + lr;
            const bool ok = (long long)tile * RB + r < p.M;
            dma16(rsA, dst + (d * NW + wave) * 1024, ok ? (unsigned)(base + r * ROWB + ((j ^ (r & 15)) << 4)) : OOB, p.nt_a != 0);
        }
    };

    auto issue_piece = [&](int tile, int stage, int zero, int d) {
        const size_t base = (size_t)tile * RB * ROWB;
        char* dst = ring + stage * STAGE;
        const int ln = lane + zero;
        const int lr = ln / UPR, j = ln % UPR;
        const int r = (d * NW + wave) * RPC + lr;
        const bool ok = (long long)tile * RB + r < p.M;
        dma16(rsA, dst + (d * NW + wave) * 1024, ok ? (unsigned)(base + r * ROWB + ((j ^ (r & 15)) << 4)) : OOB, p.nt_a != 0);
    };

    // fragment reads: row block i of this wave = rows (rw * TM + i) * 32 + (lane & 31) of the step tile
    int a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i] = ((rw * TM + i) * 32 + (lane & 31)) * ROWB;
    const int rsw = lane & 15;  // (row & 15) of this lane's rows: the tile's row blocks start at multiples of 32

    v16f acc[TM];
    // FUSE == 4: the waves of one SIMD (w and w + 4) leave every barrier together and would issue their DMA, run their MFMAs and
    // scan their accumulators in phase - the matrix pipe idle while both do something else.  The upper four waves therefore
    // scan the accumulators of step t AFTER the barrier of step t + 1, under the MFMAs of the lower four
    const bool late = FUSE == 4 && (p.flags & 2) != 0 && wave >= 4;
    auto filter_epi = [&](int tt) {
        if constexpr (FUSE == 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= unscale;
            const long long row0 = (long long)tt * RB + rw * TM * 32;
            // top-k admission filter: rows beyond M (a ragged last tile read zeros) are taken out first; then ONE comparison
            // of the lane's maximum tells whether any of its TM * 16 elements is a candidate (~k / Gc of them are)
            const long long rb = row0 + 4 * khalf;
            if ((long long)(tt + 1) * RB > p.M) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (rb + i * 32 + (r & 3) + 8 * (r >> 2) >= p.M) acc[i][r] = -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[i][r]);
            // Candidates are STAGED in a wave-private LDS list (an LDS atomic answers in ~100 cycles) and leave for the
            // queries' lists 64 at a time.  Appending each one through its query's global counter made the wave wait ~1 us
            // for every returning atomic, ~5 times per step of 6.6 us at two waves per SIMD: 46 % of the wave cycles parked
            // (profiles/r05e_pmc_sq_retrieval_before_staging.txt)
            if (__ballot(mx >= f_thr) != 0ull) {
                if (mx >= f_thr) {
                    const int col = n0 + (lane & 31);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[i][r];
                            if (v >= f_thr) {
                                const int row = (int)(rb + i * 32 + (r & 3) + 8 * (r >> 2)) + p.filt.col0;
                                const unsigned pos = lds_add_rtn(f_wcnt, 1u);
                                if (pos < (unsigned)FCAP) {
                                    lds_store1u(f_wbuf + 12u * pos, __float_as_uint(v));
                                    lds_store1u(f_wbuf + 12u * pos + 4u, (unsigned)row);
                                    lds_store1u(f_wbuf + 12u * pos + 8u, (unsigned)col);
                                } else {  // (staging list full - more than FCAP - 64 candidates in one step: straight to the list)
                                    filter_append(p.filt, col, v, row);
                                }
                            }
                        }
                }
                const unsigned staged = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_load1(f_wcnt));
                if (staged >= 64u) flush_staged(staged);
            }
        }
    };
    // Everything fetched so far (the filter panel, the epilogue coefficients) is waited for HERE, by an instruction the compiler
    // accounts for.  The waits of the loop are inline asm it does not see: left to itself it keeps the panel's K / 8 loads
    // "possibly in flight" around the back edge and guards the MFMAs with a ladder s_waitcnt vmcnt(K / 8 - 3) ... vmcnt(0) -
    // which in steady state waits for the NEXT tile's DMA (and this step's stores) before the step's last MFMAs
    // (the empty asm statements pin the loads above the wait: the scheduler otherwise sinks them below it)
#pragma unroll
    for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) asm volatile("" : "+v"(bf[g][ks][pl]));
    if constexpr (FUSE == 1 || FUSE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(fsc[j]), "+v"(fsh[j]));
    }
    if constexpr (FUSE == 4) asm volatile("" : "+v"(f_thr));
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
    int t = worker;
    if (t < p.tiles) issue(t, 0, 0);
    if (NS == 3 && t + p.workers < p.tiles) issue(t + p.workers, 1, 0);
    int stage = 0;
    bool first = true;
    for (; t < p.tiles; t += p.workers, stage = (NS == 2 ? stage ^ 1 : (stage == NS - 1 ? 0 : stage + 1))) {
        // the DMA of this tile was issued before this wave's stores of the previous step: vmcnt retires in order
        // (the counter has 6 bits: with 64 stores per step the oldest two are waited for as well)
        // FUSE: at least TM * 4 P16 stores (+ TM * 16 stores of the raw output, FUSE == 2) follow the DMA - a count below the
        // real one (the optional mask stores) only waits for a few of the oldest stores as well
        constexpr int NFU = TM * 4 + (FUSE == 2 ? TM * 16 : 0);
        constexpr int NST = TM * 16 + 1 > 63 ? 63 : TM * 16 + 1, NSC = FUSE ? NFU : (TM * 16 > 63 ? 63 : TM * 16);
        static_assert(NFU <= 63, "s_waitcnt vmcnt has six bits");
        if constexpr (NS == 3) {
            // younger than this tile's DMA: the next tile's DMA, if there is one, and at most two partial stores (one counted)
            if (t + p.workers < p.tiles) {
                if (first) wait_vm<DPW>();
                else wait_vm<DPW + 1>();
            } else {
                wait_vm<0>();
            }
        } else if constexpr (FUSE == 4) {
            wait_vm<0>();  // (the only vector-memory traffic besides this tile's DMA: the rare candidate appends of the last step)
        } else {
            if (first) wait_vm<0>();
            else if (p.stats != nullptr && RW == 1) wait_vm<NST>();
            else wait_vm<NSC>();
        }
        lds_barrier();
        if constexpr (FUSE == 4) {
            if (late && !first) filter_epi(t - p.workers);
        }
        const int tn = t + (NS - 1) * p.workers;
        // RW > 1: the previous step's BatchNorm partials, merged by the first lanes of the workgroup (one column each)
        if (RW > 1 && p.stats != nullptr && !first && tid < CW * 32) {
            const float4* src = sstat + (stage ^ 1) * NW * 32;
            const int c2 = tid >> 5, n = tid & 31;
            const long long row0 = (long long)(t - p.workers) * RB;
            float cnt = 0.f, mean = 0.f, m2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const long long left = p.M - (row0 + (long long)k * TM * 32);
                const float nb = (float)(left < TM * 32 ? (left > 0 ? left : 0) : TM * 32);
                if (nb > 0.f) {
                    const float4 v = src[(k * CW + c2) * 32 + n];
                    const float nt = cnt + nb, d = v.x - mean;
                    mean += d * (nb / nt);
                    m2 += v.y + d * d * (cnt * nb / nt);
                    cnt = nt;
                    lo = fminf(lo, v.z);
                    hi = fmaxf(hi, v.w);
                }
            }
            const int col = panel * (CW * 32) + tid;
            if (col < p.N) reinterpret_cast<float4*>(p.stats)[(long long)(t - p.workers) * p.N + col] = make_float4(mean, m2, lo, hi);
        }
        // (the partial store above is OLDER than everything issued below: the counted waits - which leave only the youngest
        // stores in flight - cover it; no drain here)
        // accumulate: the old C values of this step, fetched under the MFMAs (issued BEFORE the DMA so that a counted
        // wait retires them without waiting for the next tile)
        float oldc[ACC ? TM : 1][16];
        if constexpr (ACC) {
            const unsigned ldcb = (unsigned)p.ldc * 4u;
            const unsigned base = col_live ? ((unsigned)t * RB + rw * TM * 32 + 4u * khalf) * ldcb + (unsigned)(n0 + (lane & 31)) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    oldc[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, base + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldcb, 0, 0));
        }
        // ... gated by a bit mask over C (C = acc + (bit ? C : 0): the ReLU mask of the block output on the way back).  Element
        // (row, col) of an [M][N] tensor with N % 256 == 0: quad q = col / 4 of 256-column group col / 256 of the row, four
        // 64-bit words per group (one per component col % 4), bit q % 64; read as the 32-bit half that holds the bit
        unsigned mw[ACC ? TM : 1][16];
        if constexpr (ACC) {
            if (p.cmask != nullptr) {
                const unsigned col = (unsigned)(n0 + (lane & 31));
                const unsigned rowb = (unsigned)p.N >> 3;  // mask bytes per row
                const unsigned mbase = col_live ? ((unsigned)t * RB + rw * TM * 32 + 4u * khalf) * rowb + (col >> 8) * 32u + (col & 3u) * 8u + ((col >> 5) & 4u) : OOB;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        mw[i][r] = __builtin_amdgcn_raw_buffer_load_b32(rsM, mbase + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * rowb, 0, 0);
            }
        }
        // FUSE: the identity branch of this step's rows, 8 channels per lane and tile half (hi and lo planes: 2 x 16 bytes),
        // fetched under the MFMAs like the old C values above
        v4u rhi[FUSE ? TM : 1][2], rlo[FUSE ? TM : 1][2];
        if constexpr (FUSE == 1 || FUSE == 2) {
            const unsigned rowb = (unsigned)p.N * 4u;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned row = (unsigned)t * RB + (unsigned)(rw * TM + i) * 32u + (unsigned)((lane + 64 * h) >> 2);
                    const bool res_live = col_live && p.res != nullptr;  // (no identity branch: out-of-range loads return zeros)
                    const unsigned off = res_live ? row * rowb + (unsigned)(n0 >> 5) * 128u + (unsigned)(lane & 3) * 16u : OOB;
                    rhi[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rsR, off, 0, 0);
                    rlo[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rsR, res_live ? off + 64u : OOB, 0, 0);
                }
        }
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const bool dma_spread = FUSE == 4 && (p.flags & 1) != 0;  // the DMA pieces among the MFMAs instead of ahead of them
        if (tn < p.tiles && !dma_spread) issue(tn, NS == 2 ? (stage ^ 1) : (stage == 0 ? NS - 1 : stage - 1), zero);

        // ---- MFMAs of this tile: per 16-deep k step 2 fragment reads per row block, 3 products
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const char* sA = ring + stage * STAGE;
        // fragment prefetch distance: 1 k step (two register sets).  (The retrieval filter runs two waves per SIMD - its 128-VGPR
        // query panel - at 44 % MFMA-busy; a distance of 2 (three sets, 256 VGPRs) measured the same 16.4 ms: TRID_TOPK_PF)
        constexpr int PF = FUSE == 4 ? TRID_TOPK_PF : 1;
        f16x8 af[PF + 1][TM][2];
        auto fetch = [&](int q, f16x8(&dst)[TM][2]) {  // q = k group * 2 + k step
            const int g = q >> 1, ks = q & 1;
            const int u0 = g * 8 + 2 * ks + khalf;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                dst[i][0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sA + a_off[i] + (((u0) ^ rsw) << 4)));
                dst[i][1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sA + a_off[i] + (((u0 + 4) ^ rsw) << 4)));
            }
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) fetch(q, af[q]);
#pragma unroll
        for (int q = 0; q < 2 * KG; ++q) {
            if (q + PF < 2 * KG) fetch(q + PF, af[(q + PF) % (PF + 1)]);
            if constexpr (FUSE == 4) {
                if (dma_spread && tn < p.tiles && (q & 1) == 0 && (q >> 1) < DPW) issue_piece(tn, stage ^ 1, zero, q >> 1);
            }
            const int g = q >> 1, ks = q & 1;
            constexpr int NB = PF + 1;
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q % NB][i][0], bf[g][ks][1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q % NB][i][1], bf[g][ks][0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[q % NB][i][0], bf[g][ks][0], acc[i], 0, 0, 0);
        }

        // ---- epilogue
        if constexpr (FUSE != 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= unscale;
        }
        if constexpr (ACC) {
            wait_vm<DPW>();  // the old C values (older than this step's DMA)
            if (p.cmask != nullptr) {
                const unsigned bit = ((unsigned)(n0 + (lane & 31)) >> 2) & 31u;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] += ((mw[i][r] >> bit) & 1u) ? oldc[i][r] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] += oldc[i][r];
            }
        }
        const long long row0 = (long long)t * RB + rw * TM * 32;  // first row of this wave's blocks
        if (p.stats != nullptr) {
            const long long left = p.M - row0;
            const int cnt_w = __builtin_amdgcn_readfirstlane((int)(left < TM * 32 ? (left > 0 ? left : 0) : TM * 32));
            float sum = 0.f, lo = INFINITY, hi = -INFINITY, m2 = 0.f, mean = 0.f;
            if (cnt_w == TM * 32) {  // (wave-uniform: every tile but a ragged last one)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sum += acc[i][r];
                        lo = fminf(lo, acc[i][r]);
                        hi = fmaxf(hi, acc[i][r]);
                    }
                sum += __shfl_xor(sum, 32, 64);
                mean = sum * (1.f / (TM * 32));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[i][r] - mean;
                        m2 += d * d;
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool in = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf < cnt_w;
                        sum += in ? acc[i][r] : 0.f;
                        lo = fminf(lo, in ? acc[i][r] : INFINITY);
                        hi = fmaxf(hi, in ? acc[i][r] : -INFINITY);
                    }
                sum += __shfl_xor(sum, 32, 64);
                mean = cnt_w > 0 ? sum / (float)cnt_w : 0.f;
                int cnt2 = cnt_w;  // (opaque copy: the 64 row predicates are re-derived, not kept alive in SGPR pairs across the loops)
                asm volatile("" : "+s"(cnt2));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool in = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf < cnt2;
                        const float d = acc[i][r] - mean;
                        m2 += in ? d * d : 0.f;
                    }
            }
            m2 += __shfl_xor(m2, 32, 64);
            lo = fminf(lo, __shfl_xor(lo, 32, 64));
            hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
            if constexpr (RW == 1) {
                // (every wave, one instruction: the counted wait at the top of the next step relies on it)
                const unsigned off = (col_live && khalf == 0 && cnt_w > 0) ? (unsigned)(((unsigned)t * (unsigned)p.N + n0 + (lane & 31)) * 16u) : OOB;
                const v4f sv = {mean, m2, lo, hi};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, sv), rsS, off, 0, 0);
            } else {
                if (khalf == 0) lds_store4(sstat + stage * NW * 32 + wave * 32 + (lane & 31), make_float4(mean, m2, lo, hi));
            }
        }
        {
            // 32-bit byte offsets (the tensor stays below 2 GB); rows >= M lie beyond the descriptor's range: dropped.  A wave
            // whose columns do not exist stores everything out of range
            // (no C - the statistics-only pass: every store is out of range, the instruction count the waits rely on stays)
            const unsigned ldcb = (unsigned)p.ldc * 4u;
            const unsigned base = (col_live && p.C != nullptr) ? (unsigned)row0 * ldcb + 4u * khalf * ldcb + (unsigned)(n0 + (lane & 31)) * 4u : OOB;
            if constexpr (FUSE != 1 && FUSE != 3 && FUSE != 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][r];
                        store_c(__float_as_uint(v), rsC, base + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldcb, p.nt_c != 0);
                    }
            }
        }
        if constexpr (FUSE == 4) {
            if (!late) filter_epi(t);
        }
        if constexpr (FUSE == 1 || FUSE == 2) {
            // out = relu(y * scale + shift + identity) as a P16 tensor, the arithmetic of bn_apply_kernel<1, 1> (bn_pool.hip).
            // The tile is transposed through a wave-private LDS tile (accumulator layout: one column per lane; P16 rows want
            // 8 consecutive channels per lane: 16 bytes of the hi plane, 16 of the lo plane).  The LDS stores are inline asm
            // (beside the LDS-DMA in flight hipcc would order a visible ds_write behind s_waitcnt vmcnt(0)) and take the
            // accumulators after a VALU instruction (the unscale above), not straight from the MFMA.
            wait_vm<DPW + (FUSE == 2 ? TM * 16 : 0)>();  // the identity branch (older than this step's DMA and C stores)
            float* const tw = tbuf + wave * (32 * TP);
            const unsigned rowb = (unsigned)p.N * 4u;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][r];
                    lds_store1(tw + ((r & 3) + 8 * (r >> 2) + 4 * khalf) * TP + (lane & 31), v);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                unsigned long long bal[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rl = (lane + 64 * h) >> 2, oc = lane & 3;
                    const float4 y0 = *reinterpret_cast<const float4*>(tw + rl * TP + 8 * oc);
                    const float4 y1 = *reinterpret_cast<const float4*>(tw + rl * TP + 8 * oc + 4);
                    const float yv[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {  // dword q of a plane = channels 2q, 2q + 1
                        const unsigned hw = rhi[i][h][q], lw = rlo[i][h][q];  // (a bit_cast of the vector ELEMENT itself reads element 0)
                        const f16x2 hh = __builtin_bit_cast(f16x2, hw), ll = __builtin_bit_cast(f16x2, lw);
                        const float r0 = ((float)hh.x + (float)ll.x) * rinv, r1 = ((float)hh.y + (float)ll.y) * rinv;
                        v[2 * q] = fmaf(yv[2 * q], fsc[2 * q], fsh[2 * q]) + r0;
                        v[2 * q + 1] = fmaf(yv[2 * q + 1], fsc[2 * q + 1], fsh[2 * q + 1]) + r1;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) bal[h][j] = __ballot(v[j] > 0.f);
                    if (p.relu) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                    }
                    v4u oh, ol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned hq, lq;
                        f16_split2(v[2 * q] * oscale, v[2 * q + 1] * oscale, hq, lq);
                        oh[q] = hq;
                        ol[q] = lq;
                    }
                    const unsigned row = (unsigned)t * RB + (unsigned)(rw * TM + i) * 32u + (unsigned)rl;
                    const unsigned off = col_live ? row * rowb + (unsigned)(n0 >> 5) * 128u + (unsigned)oc * 16u : OOB;
                    if (p.ev.out_tmax != nullptr && col_live && row < (unsigned)p.M) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned a = __builtin_bit_cast(unsigned, v[j]) & 0x7fffffffu;
                            tmax_bits = a > tmax_bits ? a : tmax_bits;
                        }
                    }
                    // Whole 128-byte lines per store instruction: the four lanes of row r (even) and the four of row r + 1 swap
                    // low planes - the first store writes row r (high parts from its own lanes, low parts through the lanes of
                    // row r + 1), the second row r + 1.  (With one row's 64-byte plane half per lane quad and instruction every
                    // store covered half lines: split_common.h p16_store4_pair.)
                    v4u pol;
#pragma unroll
                    for (int q = 0; q < 4; ++q) pol[q] = (unsigned)__shfl_xor((int)ol[q], 4, 64);
                    const bool odd_row = (rl & 1) != 0;
                    const unsigned off_e = col_live ? (row & ~1u) * rowb + (unsigned)(n0 >> 5) * 128u + (unsigned)oc * 16u + (odd_row ? 64u : 0u) : OOB;
                    const unsigned off_o = col_live ? (row | 1u) * rowb + (unsigned)(n0 >> 5) * 128u + (unsigned)oc * 16u + (odd_row ? 0u : 64u) : OOB;
                    const v4u d_e = odd_row ? pol : oh, d_o = odd_row ? oh : pol;
                    if (p.nt_o) {
                        __builtin_amdgcn_raw_buffer_store_b128(d_e, rsO, off_e, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b128(d_o, rsO, off_o, 0, 2);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(d_e, rsO, off_e, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(d_o, rsO, off_o, 0, 0);
                    }
                }
                if (p.mask != nullptr) {
                    // ReLU bits in bn_apply's layout: quad q of a 256-column group -> bit q of the word of its component; this
                    // wave's 8 quads of a row are byte (n0 / 32) % 8 of the row's four words.  Lane L writes the byte of
                    // (row L / 4 of the half, component L % 4): from the two ballots that hold that component (channel j = k
                    // of an octet is quad 2 oc, j = k + 4 quad 2 oc + 1), 4 octet bits of the row each, interleaved
                    const int rr = lane >> 2, k = lane & 3;
                    const unsigned cq0 = (unsigned)(n0 >> 2);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned long long b0 = k == 0 ? bal[h][0] : k == 1 ? bal[h][1] : k == 2 ? bal[h][2] : bal[h][3];
                        const unsigned long long b1 = k == 0 ? bal[h][4] : k == 1 ? bal[h][5] : k == 2 ? bal[h][6] : bal[h][7];
                        const unsigned n0b = (unsigned)(b0 >> (4 * rr)) & 15u, n1b = (unsigned)(b1 >> (4 * rr)) & 15u;
                        const unsigned e0 = (n0b & 1u) | ((n0b & 2u) << 1) | ((n0b & 4u) << 2) | ((n0b & 8u) << 3);
                        const unsigned e1 = (n1b & 1u) | ((n1b & 2u) << 1) | ((n1b & 4u) << 2) | ((n1b & 8u) << 3);
                        const unsigned byte = e0 | (e1 << 1);
                        const unsigned row = (unsigned)t * RB + (unsigned)(rw * TM + i) * 32u + (unsigned)(16 * h + rr);
                        const unsigned grp = row * ((unsigned)p.N >> 8) + (cq0 >> 6);
                        const unsigned off = col_live ? grp * 32u + (unsigned)k * 8u + ((cq0 & 63u) >> 3) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)byte, rsK, off, 0, 0);
                    }
                }
            }
        }
        first = false;
    }
    if constexpr (FUSE == 4) {
        if (late && !first) filter_epi(t - p.workers);
        const unsigned staged = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_load1(f_wcnt));
        if (staged != 0u) flush_staged(staged);
    }
    if constexpr (FUSE == 1 || FUSE == 2) {
        if (p.ev.out_tmax != nullptr) {  // one atomic per workgroup
            tmax_bits = wave_umax(tmax_bits);
            lds_barrier();
            unsigned* redu = reinterpret_cast<unsigned*>(sstat);
            if (lane == 0) redu[wave] = tmax_bits;
            __syncthreads();
            if (tid == 0) {
                unsigned r = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) r = redu[w] > r ? redu[w] : r;
                if (r != 0) atomicMax(reinterpret_cast<unsigned*>(p.ev.out_tmax), r);
            }
        }
    }
    if (RW > 1 && p.stats != nullptr && !first) {  // the last step's partials
        lds_barrier();
        if (tid < CW * 32) {
            const int tl = t - p.workers;
            const float4* src = sstat + (stage ^ 1) * NW * 32;
            const int c2 = tid >> 5, n = tid & 31;
            const long long row0 = (long long)tl * RB;
            float cnt = 0.f, mean = 0.f, m2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                const long long left = p.M - (row0 + (long long)k * TM * 32);
                const float nb = (float)(left < TM * 32 ? (left > 0 ? left : 0) : TM * 32);
                if (nb > 0.f) {
                    const float4 v = src[(k * CW + c2) * 32 + n];
                    const float nt = cnt + nb, d = v.x - mean;
                    mean += d * (nb / nt);
                    m2 += v.y + d * d * (cnt * nb / nt);
                    cnt = nt;
                    lo = fminf(lo, v.z);
                    hi = fmaxf(hi, v.w);
                }
            }
            const int col = panel * (CW * 32) + tid;
            if (col < p.N) reinterpret_cast<float4*>(p.stats)[(long long)tl * p.N + col] = make_float4(mean, m2, lo, hi);
        }
    }
}

template <int K, int CW, int TM, bool ACC, int FUSE = 0>
static int launch_stream(StreamParams& p, hipStream_t stream) {
    constexpr int RW = 8 / CW, RB = RW * TM * 32;
    constexpr bool FEPI = FUSE == 1 || FUSE == 2;
    const size_t lds = (size_t)(FUSE == 3 ? 3 : 2) * RB * K * 4 + ((RW > 1 || FEPI) ? 2 * 8 * 32 * sizeof(float4) : 0) + (FEPI ? (size_t)8 * 32 * 36 * 4 : 0);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)gemm_p16_stream_kernel<K, CW, TM, ACC, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_gemm_p16_stream: cannot reserve LDS: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    p.panels = (p.N + CW * 32 - 1) / (CW * 32);
    p.tiles = (p.M + RB - 1) / RB;
    int per = 256 / p.panels;          // persistent workgroups per panel: one per CU in all
    per = per / 8 * 8;                 // whole XCD rounds
    if (per < 8) per = 8;
    int need = (p.tiles + 7) / 8 * 8;  // no more workers than tiles
    p.workers = std::min(per, need);
    const int grid = p.workers * p.panels;
    hipLaunchKernelGGL((gemm_p16_stream_kernel<K, CW, TM, ACC, FUSE>), dim3(grid), dim3(512), lds, stream, p);
    return check_launch("trid_gemm_p16_stream");
}

// shapes -> (column waves, row blocks per wave): two stages of RB x K x 4 bytes must fit the LDS, so K = 256 steps 64 rows
template <int K>
static int pick_stream(StreamParams& p, hipStream_t stream) {
    const bool acc = p.accumulate != 0;
    if constexpr (K == 256) {  // (beside the 128-VGPR filter panel the old C values fit for ONE row block per wave and step)
        if (p.N <= 128) return TRID_E_UNSUPPORTED;
        if (acc) return launch_stream<K, 8, 1, true>(p, stream);
        if (p.C == nullptr) return launch_stream<K, 8, 1, false, 3>(p, stream);  // statistics only: three stages of 32 rows
        return launch_stream<K, 8, 2, false>(p, stream);
    } else {
        if (p.N > 128 && !acc && p.C == nullptr) {  // statistics only: three stages (K = 128: 64-row steps, 3 x 32 KB)
            if constexpr (K == 64) return launch_stream<K, 8, 4, false, 3>(p, stream);
            else return launch_stream<K, 8, 2, false, 3>(p, stream);
        }
        if (p.N > 128) return acc ? launch_stream<K, 8, 2, true>(p, stream) : launch_stream<K, 8, 4, false>(p, stream);
        if (p.N > 64) return acc ? launch_stream<K, 4, 2, true>(p, stream) : launch_stream<K, 4, 2, false>(p, stream);
        return acc ? launch_stream<K, 2, 1, true>(p, stream) : launch_stream<K, 2, 1, false>(p, stream);
    }
}

}  // namespace trid

using namespace trid;

// rows per BatchNorm partial (= rows per step) of the streaming kernel for this shape, 0 when it does not apply
extern "C" int trid_gemm_p16_stream_rows(int M, int N, int K, int accumulate) {
    if (!(K == 64 || K == 128 || K == 256) || N <= 0 || N % 32 != 0 || M <= 0) return 0;
    // (31-bit buffer offsets: operands of 2 GB and more - per-GPU batches beyond ~640 images at 384 x 128 - stay on the tile kernel)
    if ((long long)M * K * 4 >= (1ll << 31) || (long long)M * N * 4 >= (1ll << 31)) return 0;
    if (K == 256 && N <= 128) return 0;  // (256 -> 128 measured no faster than the tile kernel: 146 vs 142 us)
    if (K == 256 && accumulate) return 32;
    if (N > 128) return (accumulate || K == 256) ? 64 : 128;
    if (N > 64) return K == 256 ? 64 : 128;
    return 128;
}

// ... of the statistics-only pass (C == NULL)
extern "C" int trid_gemm_p16_stream_stats_rows(int M, int N, int K) {
    if (N > 128 && K == 128) return 64;
    if (N > 128 && K == 256) return 32;
    return trid_gemm_p16_stream_rows(M, N, K, 0);
}

extern "C" int trid_gemm_p16_stream(const void* A, const float* a_amax, const void* B, const float* b_amax, float* C, long long ldc,
                                    float* stats, int M, int N, int K, int accumulate, const uint64_t* c_mask, void* stream) {
    TRID_REQUIRE(A && B && (C || (stats && !accumulate)) && a_amax && b_amax, "trid_gemm_p16_stream: null operand (C may be null only for a statistics-only pass)");
    TRID_REQUIRE(trid_gemm_p16_stream_rows(M, N, K, accumulate) > 0, "trid_gemm_p16_stream: needs K in {64, 128, 256} and N %% 32 == 0 (M=%d N=%d K=%d)", M, N, K);
    TRID_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && (!stats || aligned16(stats)), "trid_gemm_p16_stream: operands must be 16-byte aligned");
    TRID_REQUIRE(ldc >= N && !(stats && accumulate), "trid_gemm_p16_stream: ldc >= N; BatchNorm partials only without accumulate");
    TRID_REQUIRE((long long)M * K * 4 < (1ll << 31) && (long long)M * ldc * 4 < (1ll << 31), "trid_gemm_p16_stream: operands must stay below 2 GB (31-bit buffer offsets)");
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const char*)A; p.B = (const char*)B; p.C = C; p.stats = stats;
    p.a_amax = a_amax; p.b_amax = b_amax;
    p.M = M; p.N = N; p.ldc = ldc; p.accumulate = accumulate;
    TRID_REQUIRE(c_mask == nullptr || (accumulate && ldc == N && N % 256 == 0), "trid_gemm_p16_stream: c_mask needs accumulate, ldc == N and N %% 256 == 0 (N=%d)", N);
    p.cmask = reinterpret_cast<const unsigned*>(c_mask);
    static const int nt_env = getenv("TRID_STREAM_NT") ? atoi(getenv("TRID_STREAM_NT")) : -1;
    // measured per shape (tools/exp/nt_ab2.sh, profiles/r04i_stream_nontemporal.txt): non-temporal OUTPUT stores pay for outputs
    // beyond the Infinity Cache (layer1 conv3: 103.6 -> 81.9 us = 6.1 TB/s); non-temporal activation loads only beside the
    // read-modify-write of a large C (the accumulate forms: 181 -> 171 us), elsewhere they cost 5-15 %
    p.nt_a = nt_env >= 0 ? (nt_env & 1) : (accumulate && (long long)M * N * 4 >= 2 * STREAM_NT_MIN_BYTES);
    p.nt_c = nt_env >= 0 ? ((nt_env >> 1) & 1) : ((long long)M * N * 4 >= STREAM_NT_MIN_BYTES && !accumulate);
    hipStream_t s = (hipStream_t)stream;
    if (K == 64) return pick_stream<64>(p, s);
    if (K == 128) return pick_stream<128>(p, s);
    return pick_stream<256>(p, s);
}

// Retrieval: candidates of Q queries (P16 [Qp][256], Qp = Q rounded up to 32, padding rows zero) against G gallery rows (P16
// [G][256]): every similarity >= thr[query * thr_stride] is appended to the query's list (value, gallery row + col0).
// 1280 persistent workgroups = 5 full rounds of the chip for Q = 1e4 (40 query panels x 32 workers; the panels of one worker
// index share an XCD and walk the same gallery tiles: the gallery crosses the fabric once per XCD, not once per panel).
int trid::stream_topk_filter(const void* g16, const float* g_amax, const void* q16, const float* q_amax, int G, int Q, const GemmFilter& filt,
                             hipStream_t stream) {
    const int Qp = (Q + 31) / 32 * 32;
    TRID_REQUIRE(g16 && q16 && g_amax && q_amax && G > 0 && Q > 0 && filt.thr && filt.cnt && filt.cand && filt.overflow, "stream_topk_filter: null operand");
    TRID_REQUIRE(aligned16(g16) && aligned16(q16) && (long long)G * 1024 < (1ll << 31), "stream_topk_filter: 16-byte aligned operands, gallery shard below 2 GB");
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const char*)g16; p.B = (const char*)q16; p.C = nullptr;
    p.a_amax = g_amax; p.b_amax = q_amax;
    p.M = G; p.N = Qp; p.ldc = Qp;
    p.filt = filt; p.n_real = Q;
    static const int f_env = getenv("TRID_TOPK_FLAGS") ? atoi(getenv("TRID_TOPK_FLAGS")) : 3;  // (A/B runs: tools/exp/r05_run9.sh)
    p.flags = f_env;
    constexpr int K = 256, CW = 8, TM = 2, RB = TM * 32;
    const size_t lds = (size_t)2 * RB * K * 4 + 2 * 8 * 32 * sizeof(float4) + (size_t)8 * 3 * 128 * 4;  // ring, counters, staging lists
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)gemm_p16_stream_kernel<K, CW, TM, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) {
        set_error("stream_topk_filter: cannot reserve LDS: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    p.panels = (Qp + CW * 32 - 1) / (CW * 32);
    p.tiles = (G + RB - 1) / RB;
    // workers per panel (a multiple of 8: one per XCD and round): the count that minimises rounds of the chip x (steps per
    // worker + the panel fetch), e.g. 32 for 40 panels (1280 workgroups = 5 full rounds) whatever the segment's length
    int workers = 8;
    {
        double best = 1e300;
        for (int w = 8; w <= 256; w += 8) {
            const long long rounds = ((long long)p.panels * w + 255) / 256;
            const double cost = (double)rounds * ((double)((p.tiles + w - 1) / w) * 5.0 + 3.0);
            if (cost < best * 0.999) { best = cost; workers = w; }
        }
    }
    static const int w_env = getenv("TRID_TOPK_WORKERS") ? atoi(getenv("TRID_TOPK_WORKERS")) : 0;  // (experiments)
    if (w_env > 0) workers = w_env;
    p.workers = std::min(workers, (p.tiles + 7) / 8 * 8);
    hipLaunchKernelGGL((gemm_p16_stream_kernel<K, CW, TM, false, 4>), dim3(p.workers * p.panels), dim3(512), lds, stream, p);
    return check_launch("stream_topk_filter");
}

// shapes the fused conv3 + BatchNorm + identity + ReLU kernel covers (K = planes of layer1 - layer3, N = 4 planes)
extern "C" int trid_conv1x1_bn_res_p16_ok(int M, int N, int K) {
    return (K == 64 || K == 128 || K == 256) && N > 0 && N % 256 == 0 && M > 0 && (long long)M * N * 4 < (1ll << 31);
}

// eval mode: out = act(scale * (A . B^T) + shift (+ res)) as a P16 tensor on the fused streaming kernel, the output's scale from
// the analytic bound (gemm_common.h EvalBound), its true maximum folded into *out_tmax
extern "C" int trid_conv1x1_eval_p16(const void* A, const float* a_amax, const void* B, const float* b_amax, const float* bn_scale,
                                     const float* bn_shift, const void* res, const float* res_amax, void* out, const float* eval_coef,
                                     const float* eval_tin, const float* eval_tres, float* out_bound, float* out_tmax, int M, int N, int K,
                                     int relu, void* stream) {
    TRID_REQUIRE(A && B && a_amax && b_amax && bn_scale && bn_shift && out && eval_coef && eval_tin && out_bound, "trid_conv1x1_eval_p16: null operand");
    TRID_REQUIRE(!res || (res_amax && eval_tres), "trid_conv1x1_eval_p16: a residual needs its amax scalar and its true maximum");
    TRID_REQUIRE(trid_conv1x1_bn_res_p16_ok(M, N, K), "trid_conv1x1_eval_p16: needs K in {64, 128, 256}, N %% 256 == 0, tensors below 2 GB (M=%d N=%d K=%d)", M, N, K);
    TRID_REQUIRE(aligned16(A) && aligned16(B) && (!res || aligned16(res)) && aligned16(out), "trid_conv1x1_eval_p16: operands must be 16-byte aligned");
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const char*)A; p.B = (const char*)B; p.C = nullptr;
    p.a_amax = a_amax; p.b_amax = b_amax;
    p.M = M; p.N = N; p.ldc = N;
    p.bn_scale = bn_scale; p.bn_shift = bn_shift;
    p.res = (const char*)res; p.res_amax = res_amax;
    p.out = (char*)out;
    p.ev.coef = eval_coef; p.ev.tin = eval_tin; p.ev.tres = res ? eval_tres : nullptr;
    p.ev.out_bound = out_bound; p.ev.out_tmax = out_tmax;
    p.relu = relu;
    static const int nt_env = getenv("TRID_STREAM_NT") ? atoi(getenv("TRID_STREAM_NT")) : -1;
    const bool big = (long long)M * N * 4 >= STREAM_NT_MIN_BYTES;
    p.nt_o = nt_env >= 0 ? ((nt_env >> 1) & 1) : big;
    hipStream_t s = (hipStream_t)stream;
    if (K == 64) return launch_stream<64, 8, 2, false, 1>(p, s);
    if (K == 128) return launch_stream<128, 8, 2, false, 1>(p, s);
    return launch_stream<256, 8, 1, false, 1>(p, s);
}

extern "C" int trid_conv1x1_bn_res_p16(const void* A, const float* a_amax, const void* B, const float* b_amax, float* y,
                                       const float* bn_scale, const float* bn_shift, const void* res, const float* res_amax,
                                       void* out, const float* bound_a, const float* bound_b, float* bound_sum,
                                       uint64_t* relu_mask, int M, int N, int K, int relu, void* stream) {
    TRID_REQUIRE(A && B && a_amax && b_amax && bn_scale && bn_shift && res && res_amax && out && bound_a, "trid_conv1x1_bn_res_p16: null operand");
    TRID_REQUIRE(trid_conv1x1_bn_res_p16_ok(M, N, K), "trid_conv1x1_bn_res_p16: needs K in {64, 128, 256}, N %% 256 == 0, tensors below 2 GB (M=%d N=%d K=%d)", M, N, K);
    TRID_REQUIRE(aligned16(A) && aligned16(B) && aligned16(res) && aligned16(out) && (!y || aligned16(y)) && (!relu_mask || aligned16(relu_mask)),
                 "trid_conv1x1_bn_res_p16: operands must be 16-byte aligned");
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const char*)A; p.B = (const char*)B; p.C = y;
    p.a_amax = a_amax; p.b_amax = b_amax;
    p.M = M; p.N = N; p.ldc = N;
    p.bn_scale = bn_scale; p.bn_shift = bn_shift;
    p.res = (const char*)res; p.res_amax = res_amax;
    p.out = (char*)out; p.oa = bound_a; p.ob = bound_b; p.osum = bound_sum;
    p.mask = reinterpret_cast<unsigned char*>(relu_mask);
    p.relu = relu;
    static const int nt_env = getenv("TRID_STREAM_NT") ? atoi(getenv("TRID_STREAM_NT")) : -1;
    const bool big = (long long)M * N * 4 >= STREAM_NT_MIN_BYTES;
    p.nt_a = 0;
    p.nt_c = nt_env >= 0 ? ((nt_env >> 1) & 1) : big;
    p.nt_o = nt_env >= 0 ? ((nt_env >> 1) & 1) : big;
    hipStream_t s = (hipStream_t)stream;
    if (K == 64) return y ? launch_stream<64, 8, 2, false, 2>(p, s) : launch_stream<64, 8, 2, false, 1>(p, s);
    if (K == 128) return y ? launch_stream<128, 8, 2, false, 2>(p, s) : launch_stream<128, 8, 2, false, 1>(p, s);
    return y ? launch_stream<256, 8, 1, false, 2>(p, s) : launch_stream<256, 8, 1, false, 1>(p, s);  // (one row block per wave: the 128-VGPR filter panel)
}
