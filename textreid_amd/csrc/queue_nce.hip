// Fused queue similarity + masked InfoNCE, forward AND gradient, in one pass over the MoCo queues.
//
// Reference: head.py:148-170 (negative filter, pos / neg logits of both modalities) followed by
// losses.py:206-217 (cross entropy over [pos | negs] / T with label 0).  The reference clones the
// queues, gathers the unfiltered columns and materialises [B, |neg|] logits twice; here the
// [K, C] row-major queues are streamed ONCE, 32 rows at a time, and nothing of size B x K exists:
//
//   per 32-row queue tile (coalesced 1 KB rows -> split into bf16 planes -> LDS, two images):
//     S^T[j, b] = <queue_j, q_b>                 MFMA 32x32x16, A = tile (LDS), B = queries (registers)
//     P[j, b]   = hit_j ? 0 : exp((S - bound)/T) in the accumulator layout, no shuffles
//     l[b]     += sum_j P[j, b]                  in-lane adds (+ one half-wave shuffle at the very end)
//     O[b, :]  += sum_j P[j, b] * queue_j        MFMA again: A = P (registers, as produced), B = tile (LDS)
//
// Queries and queue rows are L2-normalised (head.py:128-129,140,145; queue columns are normalised keys),
// so every logit lies in [-bound/T, bound/T] with bound = 1 and a FIXED shift replaces the running
// maximum of an online softmax: exp((s - 1)/T) is in [e^-28.6, 1] at T = 0.07 - no rescaling of O, no
// overflow, no underflow.  Each workgroup owns a contiguous range of queue rows for 128 queries (4 waves
// x 32) and writes its partial (l, O); a small pre-pass writes the batch-wide filter flags (one byte per queue row); a finish kernel folds the partials in a fixed order (bit-
// reproducible, no atomics), adds the positive pair <q_b, key_b> and emits the per-row loss and dL/dq.
//
// Arithmetic: fp32-class.  Default (precision 6): queue_nce_f16_kernel below - fp16 two-plane split with fixed
// scales, 3 + 3 MFMA products per (query, row, channel), hand-pipelined.  precision = 3: fp32 operands split on
// the fly into three bf16 planes exactly as in gemm_bf16.hip; S uses the six significant plane products (dropped
// terms <= 2^-26), P (two planes, relative error 2^-17) times the three queue planes uses five.  precision = 1
// keeps one bf16 plane of everything (bf16-autocast arithmetic, configs[3]).

#include <mutex>
#include <type_traits>

#include "split_common.h"

namespace trid {

constexpr int QC = 256;          // embedding width this kernel is built for (FEATURE_SIZE of the MoCo configs)
constexpr int QCH = QC / 16;     // 16-wide K chunks of the similarity product
constexpr int QCT = QC / 32;     // 32-wide column tiles of O
constexpr int QTILE = 32;        // queue rows per tile
constexpr int QB = 128;          // queries per workgroup (4 waves x 32)
constexpr int QMAXB = 8192;      // largest (global) batch (limit of trid_queue_hit_mask)
constexpr int A1_SLOTS = 2 * QCH * 33;  // [chunk*2 + half][33] 16-byte slots (row j at +j)
constexpr int A2_SLOTS = 4 * QC;        // [kk*2 + half][c]
__host__ __device__ constexpr int a2_sw(int c) { return c ^ ((c >> 3) & 3); }

struct QnceParams {
    const float* q[2];       // [B, QC] normalised queries: modality 0 = image queries, 1 = text queries
    const float* queue[2];   // [K, QC] row-major: modality 0 reads the TEXT queue, 1 the IMAGE queue
    const uint8_t* hit;      // [K] batch-wide same-id flags: queue row k is filtered when id_queue[k] equals ANY id of the batch
    const long long* id_queue;  // [K] and
    const long long* ids;       // [B]: the hashed in-kernel filter of queue_nce_f16_kernel<true> (B <= F_HASH_MAXB) reads these instead
    float* part_l;           // [2][nwg][Bp]
    float* part_o;           // [2][nwg][Bp][QC]
    int B, Bp, K, tiles_per_wg, nwg;
    float c1, c0;            // P = exp2(s * c1 + c0)
};

template <int NPL>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[NPL]) {
    unsigned w[NPL][4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        float r0 = v[2 * qd], r1 = v[2 * qd + 1];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const unsigned h = cvt_pk_bf16(r0, r1);
            w[pl][qd] = h;
            if (pl + 1 < NPL) {
                r0 -= __builtin_bit_cast(float, h << 16);
                r1 -= __builtin_bit_cast(float, h & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) out[pl] = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

// rows of a 32-row tile owned by (kk, h): exactly the queue rows the accumulator layout of a 32x32 MFMA
// puts in registers 8kk..8kk+7 of the lanes with (lane >> 5) == h
__device__ __forceinline__ int tile_row(int kk, int h, int i) { return 16 * kk + 4 * h + (i & 3) + 8 * (i >> 2); }

template <int NPL>
__global__ __launch_bounds__(256, 1) void queue_nce_kernel(QnceParams p) {
    constexpr int NPP = NPL == 3 ? 2 : 1;  // planes of P
    extern __shared__ __attribute__((aligned(16))) uint4 qsm[];
    uint4* A1 = qsm;                      // [NPL][A1_SLOTS]
    uint4* A2 = qsm + NPL * A1_SLOTS;     // [NPL][A2_SLOTS], column c at slot sw(c): conflict-free writes AND reads

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wg = blockIdx.x, bb = blockIdx.y, mod = blockIdx.z;
    const float* __restrict__ Q = p.q[mod];
    const float* __restrict__ Kq = p.queue[mod];

    // ---- this wave's 32 queries as B-operand fragments (registers, all planes)
    bf16x8 qf[NPL][QCH];
    {
        const int b = bb * QB + wave * 32 + l31;
        const float* src = Q + (long long)(b < p.B ? b : 0) * QC + 8 * half;
#pragma unroll
        for (int ch = 0; ch < QCH; ++ch) {
            float4 u = *reinterpret_cast<const float4*>(src + 16 * ch);
            float4 v = *reinterpret_cast<const float4*>(src + 16 * ch + 4);
            if (b >= p.B) u = v = make_float4(0.f, 0.f, 0.f, 0.f);
            const float vals[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            uint4 pl[NPL];
            split8<NPL>(vals, pl);
#pragma unroll
            for (int k = 0; k < NPL; ++k) qf[k][ch] = __builtin_bit_cast(bf16x8, pl[k]);
        }
    }

    v16f O[QCT];
#pragma unroll
    for (int ct = 0; ct < QCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
    float lsum = 0.f;

    const int t_begin = wg * p.tiles_per_wg;
    const int t_end = min(p.K / QTILE, t_begin + p.tiles_per_wg);
    // loader role of this thread: wave -> (kk, h) row set, lane -> 4 consecutive columns
    const int lkk = wave >> 1, lh = wave & 1;
    float4 g[8];
    unsigned hpre = 0;  // this lane's dword of the prefetched tile's filter flags
    auto load_tile = [&](int t) {
        hpre = reinterpret_cast<const unsigned*>(p.hit + (long long)t * QTILE)[lane & 7];
        const float* base = Kq + (long long)t * QTILE * QC + 4 * lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const float4*>(base + (long long)tile_row(lkk, lh, i) * QC);
    };
    if (t_begin < t_end) load_tile(t_begin);

    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();  // every wave is done with the previous tile's LDS images
        {   // ---- split the prefetched rows into planes and write both LDS images
            unsigned w[8][NPL][2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float r0 = g[i].x, r1 = g[i].y, r2 = g[i].z, r3 = g[i].w;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const unsigned h0 = cvt_pk_bf16(r0, r1), h1 = cvt_pk_bf16(r2, r3);
                    w[i][pl][0] = h0;
                    w[i][pl][1] = h1;
                    if (pl + 1 < NPL) {
                        r0 -= __builtin_bit_cast(float, h0 << 16);
                        r1 -= __builtin_bit_cast(float, h0 & 0xffff0000u);
                        r2 -= __builtin_bit_cast(float, h1 << 16);
                        r3 -= __builtin_bit_cast(float, h1 & 0xffff0000u);
                    }
                }
            }
            // image 1 (A operand of S^T): slot [(c >> 3)][row], this lane holds columns 4*lane .. +3 = half a slot
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint2* dst = reinterpret_cast<uint2*>(A1 + pl * A1_SLOTS + (lane >> 1) * 33 + tile_row(lkk, lh, i)) + (lane & 1);
                    *dst = make_uint2(w[i][pl][0], w[i][pl][1]);
                }
            // image 2 (B operand of O): slot [kk*2 + h][c] = the 8 rows of this wave's row set for one column
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    unsigned d[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const unsigned a = w[2 * m][pl][cc >> 1], b = w[2 * m + 1][pl][cc >> 1];
                        d[m] = (cc & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                    }
                    A2[pl * A2_SLOTS + (lkk * 2 + lh) * QC + a2_sw(4 * lane + cc)] = make_uint4(d[0], d[1], d[2], d[3]);
                }
        }
        __syncthreads();
        // ---- batch-wide negative filter of this tile (bit j set <=> queue row j carries an id of the batch): the
        // flags of rows 4*(lane&7) .. +3 sit in this lane's prefetched dword -> one bit per row, OR over 8 lanes
        unsigned hmask = ((hpre & 0xffu) ? 1u : 0u) | ((hpre & 0xff00u) ? 2u : 0u) | ((hpre & 0xff0000u) ? 4u : 0u) | ((hpre & 0xff000000u) ? 8u : 0u);
        hmask <<= 4 * (lane & 7);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) hmask |= __shfl_xor(hmask, o, 64);
        if (t + 1 < t_end) load_tile(t + 1);  // in flight underneath the MFMA phases below

        // ---- S^T = tile . Q^T  (rows = queue rows of the tile, columns = this wave's queries).  Two accumulators
        // (even / odd chunks): with ONE wave per SIMD nothing else hides a dependent-accumulator bubble.
        v16f s, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s1[r] = 0.f;
        constexpr int NTERM = (NPL == 3) ? 6 : 1;
        constexpr int TA[6] = {1, 0, 2, 0, 1, 0};
        constexpr int TB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int ch = 0; ch < QCH; ch += 2) {
            bf16x8 a[NPL], a1[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                a[pl] = __builtin_bit_cast(bf16x8, A1[pl * A1_SLOTS + (ch * 2 + half) * 33 + l31]);
                a1[pl] = __builtin_bit_cast(bf16x8, A1[pl * A1_SLOTS + (ch * 2 + 2 + half) * 33 + l31]);
            }
#pragma unroll
            for (int k = 6 - NTERM; k < 6; ++k) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[k] < NPL ? TA[k] : 0], qf[TB[k] < NPL ? TB[k] : 0][ch], s, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[TA[k] < NPL ? TA[k] : 0], qf[TB[k] < NPL ? TB[k] : 0][ch + 1], s1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += s1[r];
        // ---- P = exp((s - bound)/T) on the unfiltered rows; row of register r: (r&3) + 8(r>>2) + 4*half
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = __builtin_amdgcn_exp2f(fmaf(s[r], p.c1, p.c0));
            pv[r] = ((hmask >> ((r & 3) + 8 * (r >> 2) + 4 * half)) & 1u) ? 0.f : x;
            lsum += pv[r];
        }
        bf16x8 pf[NPP][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const float vals[8] = {pv[8 * kk], pv[8 * kk + 1], pv[8 * kk + 2], pv[8 * kk + 3],
                                   pv[8 * kk + 4], pv[8 * kk + 5], pv[8 * kk + 6], pv[8 * kk + 7]};
            uint4 pl[NPP];
            split8<NPP>(vals, pl);
#pragma unroll
            for (int k = 0; k < NPP; ++k) pf[k][kk] = __builtin_bit_cast(bf16x8, pl[k]);
        }
        // ---- O[b, c] += sum_j P[j, b] * queue[j, c]: A = P (k-slot i of (kk, half) = tile_row(kk, half, i)), B = image 2
        constexpr int OTERM = (NPL == 3) ? 5 : 1;
        constexpr int OP[5] = {1, 0, 1, 0, 0};  // plane of P   (mm, hl, mh, hm, hh: smallest first)
        constexpr int OK[5] = {1, 2, 0, 1, 0};  // plane of the queue tile
#pragma unroll
        for (int ct = 0; ct < QCT; ct += 2)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 b[NPL], b1[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    b[pl] = __builtin_bit_cast(bf16x8, A2[pl * A2_SLOTS + (kk * 2 + half) * QC + a2_sw(ct * 32 + l31)]);
                    b1[pl] = __builtin_bit_cast(bf16x8, A2[pl * A2_SLOTS + (kk * 2 + half) * QC + a2_sw(ct * 32 + 32 + l31)]);
                }
#pragma unroll
                for (int k = 5 - OTERM; k < 5; ++k) {  // alternate the two accumulators
                    O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[OP[k] < NPP ? OP[k] : 0][kk], b[OK[k] < NPL ? OK[k] : 0], O[ct], 0, 0, 0);
                    O[ct + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[OP[k] < NPP ? OP[k] : 0][kk], b1[OK[k] < NPL ? OK[k] : 0], O[ct + 1], 0, 0, 0);
                }
            }
    }

    // ---- partials of this workgroup's queue range
    const long long slab = ((long long)mod * p.nwg + wg) * p.Bp + bb * QB + wave * 32;
    const float l = lsum + __shfl_xor(lsum, 32, 64);
    if (half == 0) p.part_l[slab + l31] = l;
#pragma unroll
    for (int ct = 0; ct < QCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            p.part_o[(slab + row) * QC + ct * 32 + l31] = O[ct][r];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32-class arithmetic on HALF the MFMA work: fp16 two-plane split with FIXED power-of-two scales.  Every
// operand is bounded by construction - queries and queue rows are unit vectors (|x| <= 1, scale 2^13), P =
// exp((s - bound)/T) lies in (0, 1] (scale 2^15) - so no amax pass is needed: x' = x * 2^e, hi = fp16(x'),
// lo = fp16(x' - hi), and a product is hi*hi + hi*lo + lo*hi in ONE fp32 accumulator (dropped lo*lo <= 2^-22),
// 3 MFMAs where the bf16 form needs 6 (S) / 5 (O).  With one wave per SIMD nothing else hides the loader, so the
// tile loop is software-pipelined by hand: two LDS buffers (2 planes x 2 images x 2 = 130 KB), the rows of tile
// t+1 are split and written to the other buffer BETWEEN the MFMAs of tile t (ONE barrier per tile), and the global
// loads of tile t+2 are issued a full tile ahead into a second register set.
constexpr int F_BUF = 2 * (A1_SLOTS + A2_SLOTS);  // 16-byte slots of one buffer: [A1 hi | A1 lo | A2 hi | A2 lo]
constexpr float F_SC = 8192.f;                    // 2^13: scale of queries and queue rows
constexpr int F_SP_LOG2 = 15;                     // P' = P * 2^15
constexpr int F_OPAD = QC + 4;                    // row pitch (floats) of the epilogue's LDS transpose
constexpr int F_HASH_SLOTS = 1024;                // open-addressing id set in LDS (8 KB behind the tile buffers)
constexpr int F_HASH_MAXB = 512;                  // ... for batches up to half its size; larger ones use the flag pre-pass
constexpr unsigned long long F_HASH_EMPTY = 0x8000000000000000ull;
__device__ __forceinline__ int id_hash(unsigned long long id) { return (int)((id * 0x9E3779B97F4A7C15ull) >> 54); }

typedef std::integral_constant<int, 0> par0;
typedef std::integral_constant<int, 1> par1;

// HASH: the batch-wide negative filter (head.py:148-157) is evaluated IN the kernel - the batch's ids go into an
// open-addressing set in LDS once per workgroup, each tile's 32 queue ids are looked up by 32 lanes (1-2 probes) and
// one ballot is the tile's mask - instead of a [K]-byte flag pre-pass (a 6 us launch in front of a 20-70 us kernel).
template <bool HASH>
__global__ __launch_bounds__(256, 1) void queue_nce_f16_kernel(QnceParams p) {
    extern __shared__ __attribute__((aligned(16))) uint4 qsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wg = blockIdx.x, bb = blockIdx.y, mod = blockIdx.z;
    const float* __restrict__ Q = p.q[mod];
    const float* __restrict__ Kq = p.queue[mod];

    // ---- this wave's 32 queries as B-operand fragments (registers, both planes)
    f16x8 qf[2][QCH];
    {
        const int b = bb * QB + wave * 32 + l31;
        const float* src = Q + (long long)(b < p.B ? b : 0) * QC + 8 * half;
#pragma unroll
        for (int ch = 0; ch < QCH; ++ch) {
            float4 u = *reinterpret_cast<const float4*>(src + 16 * ch);
            float4 v = *reinterpret_cast<const float4*>(src + 16 * ch + 4);
            if (b >= p.B) u = v = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned h[4], l[4];
            f16_split2(u.x * F_SC, u.y * F_SC, h[0], l[0]);
            f16_split2(u.z * F_SC, u.w * F_SC, h[1], l[1]);
            f16_split2(v.x * F_SC, v.y * F_SC, h[2], l[2]);
            f16_split2(v.z * F_SC, v.w * F_SC, h[3], l[3]);
            qf[0][ch] = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
            qf[1][ch] = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
        }
    }

    v16f O[QCT];
#pragma unroll
    for (int ct = 0; ct < QCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
    float lsum = 0.f;

    const int t_begin = wg * p.tiles_per_wg;
    const int t_end = min(p.K / QTILE, t_begin + p.tiles_per_wg);
    const int lkk = wave >> 1, lh = wave & 1;  // loader role: wave -> (kk, h) row set, lane -> 4 consecutive columns
    float4 g[8];         // the next tile to stage (loaded one O phase + half an S phase ahead of its first use)
    unsigned hpre = 0;   // this lane's dword of its filter flags (flag path)
    unsigned long long idpre = 0;  // lanes 0..31: the id of queue row (tile, lane) (hash path)
    unsigned w[8][2][2]; // staged tile: [row][plane][column pair]
    unsigned long long* idset = reinterpret_cast<unsigned long long*>(qsm + 2 * F_BUF);  // [F_HASH_SLOTS] + [1] sentinel flag
    if (HASH) {
        for (int i = tid; i <= F_HASH_SLOTS; i += 256) idset[i] = i < F_HASH_SLOTS ? F_HASH_EMPTY : 0ull;
        __syncthreads();
        for (int i = tid; i < p.B; i += 256) {
            const unsigned long long id = (unsigned long long)p.ids[i];
            if (id == F_HASH_EMPTY) { idset[F_HASH_SLOTS] = 1ull; continue; }  // the one value the table cannot hold
            int sl = id_hash(id);
            while (true) {
                const unsigned long long old = atomicCAS(&idset[sl], F_HASH_EMPTY, id);
                if (old == F_HASH_EMPTY || old == id) break;
                sl = (sl + 1) & (F_HASH_SLOTS - 1);
            }
        }
        __syncthreads();
    }
    auto load_tile = [&](int t) {
        t = min(t, t_end - 1);  // past the end: a harmless reload of the last tile (never consumed)
        if (HASH) idpre = (unsigned long long)p.id_queue[(long long)t * QTILE + l31];
        else hpre = reinterpret_cast<const unsigned*>(p.hit + (long long)t * QTILE)[lane & 7];
        const float* base = Kq + (long long)t * QTILE * QC + 4 * lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const float4*>(base + (long long)tile_row(lkk, lh, i) * QC);
    };
    // batch-wide negative filter of the tile in `g` (bit j set <=> queue row j carries an id of the batch).  Hash path:
    // lane j < 32 looks row j's id up in the LDS set, one ballot is the mask.  Flag path: the flags of rows
    // 4*(lane&7) .. +3 sit in this lane's prefetched dword -> one bit per row, OR over 8 lanes
    auto tile_mask = [&]() {
        if (HASH) {
            bool hit = false;
            if (idpre == F_HASH_EMPTY) {
                hit = idset[F_HASH_SLOTS] != 0ull;
            } else {
                int sl = id_hash(idpre);
                while (true) {
                    const unsigned long long e = idset[sl];
                    if (e == idpre) { hit = true; break; }
                    if (e == F_HASH_EMPTY) break;
                    sl = (sl + 1) & (F_HASH_SLOTS - 1);
                }
            }
            return (unsigned)(__ballot(hit && lane < 32) & 0xffffffffull);
        }
        unsigned m = ((hpre & 0xffu) ? 1u : 0u) | ((hpre & 0xff00u) ? 2u : 0u) | ((hpre & 0xff0000u) ? 4u : 0u) | ((hpre & 0xff000000u) ? 8u : 0u);
        m <<= 4 * (lane & 7);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) m |= __shfl_xor(m, o, 64);
        return m;
    };
    // split row i of the register tile into the two planes and write image 1 (A operand of S^T):
    // slot [(c >> 3)][row]; this lane holds columns 4*lane .. +3 = half a slot
    auto stage_row = [&](uint4* buf, int i) {
        f16_split2(g[i].x * F_SC, g[i].y * F_SC, w[i][0][0], w[i][1][0]);
        f16_split2(g[i].z * F_SC, g[i].w * F_SC, w[i][0][1], w[i][1][1]);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint2* dst = reinterpret_cast<uint2*>(buf + pl * A1_SLOTS + (lane >> 1) * 33 + tile_row(lkk, lh, i)) + (lane & 1);
            *dst = make_uint2(w[i][pl][0], w[i][pl][1]);
        }
    };
    // image 2 (B operand of O): slot [kk*2 + h][c] = the 8 rows of this wave's row set for column c = 4*lane + cc
    auto stage_col = [&](uint4* buf, int pl, int cc) {
        unsigned d[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned a = w[2 * m][pl][cc >> 1], b = w[2 * m + 1][pl][cc >> 1];
            d[m] = (cc & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        }
        buf[2 * A1_SLOTS + pl * A2_SLOTS + (lkk * 2 + lh) * QC + a2_sw(4 * lane + cc)] = make_uint4(d[0], d[1], d[2], d[3]);
    };

    const float c1 = p.c1 * (1.f / (F_SC * F_SC));
    const float c0 = p.c0 + (float)F_SP_LOG2;
    unsigned hmask_next = 0;
    auto tile = [&](int par, int t) {
        const uint4* cur = qsm + par * F_BUF;
        uint4* nxt = qsm + (par ^ 1) * F_BUF;
        __syncthreads();  // `cur` is complete, and every wave is done reading `nxt` (tile t-1)
        const unsigned hmask = hmask_next;

        // ---- S^T = tile . Q^T (rows = queue rows, columns = this wave's queries), two accumulators (even / odd
        // chunks), fragments fetched one group ahead; tile t+1 (in `g`) is split and its image 1 written between
        // the MFMAs of the second half
        v16f s, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s1[r] = 0.f;
        f16x8 a[2][2], a1[2][2];
        auto fetch_a = [&](int slot, int ch) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                a[slot][pl] = __builtin_bit_cast(f16x8, cur[pl * A1_SLOTS + (ch * 2 + half) * 33 + l31]);
                a1[slot][pl] = __builtin_bit_cast(f16x8, cur[pl * A1_SLOTS + (ch * 2 + 2 + half) * 33 + l31]);
            }
        };
        fetch_a(0, 0);
#pragma unroll
        for (int ch = 0; ch < QCH; ch += 2) {
            const int sl = (ch >> 1) & 1;
            if (ch + 2 < QCH) fetch_a(sl ^ 1, ch + 2);
            // (inline asm: the low plane of the queries lives in AGPRs - MFMA reads them there directly - which is
            // what leaves the compiler enough VGPRs to fetch fragments ahead; accumulate chains need no wait states)
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sl][1], qf[0][ch], s, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[sl][1], qf[0][ch + 1], s1, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sl][0], qf[1][ch], s, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[sl][0], qf[1][ch + 1], s1, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[sl][0], qf[0][ch], s, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[sl][0], qf[0][ch + 1], s1, 0, 0, 0);
            if (ch >= QCH / 2) {
                stage_row(nxt, ch - QCH / 2);
                stage_row(nxt, ch - QCH / 2 + 1);
            }
        }
        hmask_next = tile_mask();
        load_tile(t + 2);  // `g` is free again: a full O phase and half an S phase before it is needed
        // ---- P' = 2^15 exp((s - bound)/T) on the unfiltered rows; row of register r: (r&3) + 8(r>>2) + 4*half
        unsigned ph[2][4], pl_[2][4];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float x0 = __builtin_amdgcn_exp2f(fmaf(s[r] + s1[r], c1, c0));
            float x1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1] + s1[r + 1], c1, c0));
            x0 = ((hmask >> ((r & 3) + 8 * (r >> 2) + 4 * half)) & 1u) ? 0.f : x0;
            x1 = ((hmask >> (((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * half)) & 1u) ? 0.f : x1;
            lsum += x0 + x1;
            f16_split2(x0, x1, ph[r >> 3][(r & 7) >> 1], pl_[r >> 3][(r & 7) >> 1]);
        }
        f16x8 pf[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            pf[0][kk] = __builtin_bit_cast(f16x8, make_uint4(ph[kk][0], ph[kk][1], ph[kk][2], ph[kk][3]));
            pf[1][kk] = __builtin_bit_cast(f16x8, make_uint4(pl_[kk][0], pl_[kk][1], pl_[kk][2], pl_[kk][3]));
        }
        // ---- O[b, c] += sum_j P[j, b] * queue[j, c]: A = P (k-slot i of (kk, half) = tile_row(kk, half, i)),
        // B = image 2 (fetched one group ahead); image 2 of tile t+1 is written between the MFMA groups
        f16x8 b[2][2], b1[2][2];
        auto fetch_b = [&](int slot, int grp) {
            const int ct = (grp >> 1) * 2, kk = grp & 1;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                b[slot][pl] = __builtin_bit_cast(f16x8, cur[2 * A1_SLOTS + pl * A2_SLOTS + (kk * 2 + half) * QC + a2_sw(ct * 32 + l31)]);
                b1[slot][pl] = __builtin_bit_cast(f16x8, cur[2 * A1_SLOTS + pl * A2_SLOTS + (kk * 2 + half) * QC + a2_sw(ct * 32 + 32 + l31)]);
            }
        };
        fetch_b(0, 0);
#pragma unroll
        for (int grp = 0; grp < QCT; ++grp) {
            const int ct = (grp >> 1) * 2, kk = grp & 1, sl = grp & 1;
            if (grp + 1 < QCT) fetch_b(sl ^ 1, grp + 1);
            O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[1][kk], b[sl][0], O[ct], 0, 0, 0);
            O[ct + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[1][kk], b1[sl][0], O[ct + 1], 0, 0, 0);
            O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[0][kk], b[sl][1], O[ct], 0, 0, 0);
            O[ct + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[0][kk], b1[sl][1], O[ct + 1], 0, 0, 0);
            O[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[0][kk], b[sl][0], O[ct], 0, 0, 0);
            O[ct + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[0][kk], b1[sl][0], O[ct + 1], 0, 0, 0);
            stage_col(nxt, grp & 1, grp >> 1);  // (plane, column) pairs (0..1, 0..3) over the eight groups
        }
    };

    if (t_begin < t_end) {
        load_tile(t_begin);
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_row(qsm, i);
#pragma unroll
        for (int k = 0; k < 8; ++k) stage_col(qsm, k & 1, k >> 1);
        hmask_next = tile_mask();
        load_tile(t_begin + 1);
        for (int t = t_begin; t < t_end; ++t) tile((t - t_begin) & 1, t);
    }

    // ---- partials of this workgroup's queue range: transpose through LDS (the tile buffers are free) so that
    // every store instruction writes one full 1 KB row
    __syncthreads();
    float* ot = reinterpret_cast<float*>(qsm) + wave * (32 * F_OPAD);
    constexpr float unscale = 1.f / (F_SC * (float)(1 << F_SP_LOG2));
#pragma unroll
    for (int ct = 0; ct < QCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2) + 4 * half) * F_OPAD + ct * 32 + l31] = O[ct][r] * unscale;
    const long long slab = ((long long)mod * p.nwg + wg) * p.Bp + bb * QB + wave * 32;
    const float l = (lsum + __shfl_xor(lsum, 32, 64)) * (1.f / (float)(1 << F_SP_LOG2));
    if (half == 0) p.part_l[slab + l31] = l;
    // (each wave reads back only what it wrote: no barrier)
#pragma unroll 8
    for (int row = 0; row < 32; ++row)
        *reinterpret_cast<float4*>(p.part_o + (slab + row) * QC + 4 * lane) = *reinterpret_cast<const float4*>(ot + row * F_OPAD + 4 * lane);
}

// One workgroup per (query row, modality): positive logit, fold of the partials, loss row and dL/dq.
// Thread (grp = tid >> 6, c4 = tid & 63): group grp folds the partials w = grp, grp + 4, ... of columns
// 4*c4 .. 4*c4+3 (float4 loads, independent accumulators), then the four groups are added in a fixed order.
// ticket / loss (both or none): the workgroup that finishes LAST (a device-scope counter that starts at zero: the caller
// hands over a fresh word of a zero-filled pool, nothing is reset) also folds the 2 B loss rows - in index order, so the
// result does not depend on which workgroup that was - into loss[0] = loss_scale * sum: the block's third launch
// (trid_sum_f32) is gone.  Hand-off: rows are published with an agent-scope release before the ticket is drawn, the last
// arriver acquires once (cdna guide G16, counter form).
__global__ __launch_bounds__(256) void queue_nce_finish_kernel(QnceParams p, const float* __restrict__ key0,
                                                               const float* __restrict__ key1, float* __restrict__ loss_rows,
                                                               float* __restrict__ dq, float invT, float shift, float gs,
                                                               unsigned* __restrict__ ticket, float* __restrict__ loss, float loss_scale) {
    __shared__ float red[8];
    __shared__ unsigned last;
    __shared__ float4 osum[4][64];
    const int b = blockIdx.x, mod = blockIdx.y, c = threadIdx.x;  // blockDim.x == QC
    const int grp = c >> 6, c4 = c & 63;
    const float* __restrict__ qr = p.q[mod] + (long long)b * QC;
    const float* __restrict__ kr = (mod == 0 ? key0 : key1) + (long long)b * QC;
    float d = wave_sum(qr[c] * kr[c]);
    if ((c & 63) == 0) red[c >> 6] = d;
    float lpart = 0.f;
    for (int w = c; w < p.nwg; w += QC) lpart += p.part_l[((long long)mod * p.nwg + w) * p.Bp + b];
    lpart = wave_sum(lpart);
    if ((c & 63) == 0) red[4 + (c >> 6)] = lpart;
    const float4* __restrict__ po = reinterpret_cast<const float4*>(p.part_o + (((long long)mod * p.nwg) * p.Bp + b) * QC) + c4;
    const long long wstride = (long long)p.Bp * (QC / 4);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int w = grp;
    for (; w + 4 < p.nwg; w += 8) {
        const float4 u = po[w * wstride], v = po[(w + 4) * wstride];
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
    }
    if (w < p.nwg) {
        const float4 u = po[w * wstride];
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
    }
    osum[grp][c4] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    const float pos = (red[0] + red[1]) + (red[2] + red[3]);
    const float lneg = (red[4] + red[5]) + (red[6] + red[7]);
    const float x0 = expf(pos * invT - shift);
    const float inv = 1.f / (lneg + x0);
    const float* of = reinterpret_cast<const float*>(&osum[0][0]);
    const float o = (of[c] + of[QC + c]) + (of[2 * QC + c] + of[3 * QC + c]);
    dq[((long long)mod * p.B + b) * QC + c] = gs * (o * inv + (x0 * inv - 1.f) * kr[c]);
    // lse - pos/T = log(ltot / x0) = log1p(lneg / x0): no cancellation between the shift and log(ltot)
    if (c == 0) loss_rows[(long long)mod * p.B + b] = log1pf(lneg / x0);
    if (ticket == nullptr) return;
    if (c == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(2 * p.B - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last == 0u) return;
    if (c == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    float acc = 0.f;
    for (int i = c; i < 2 * p.B; i += QC) acc += __hip_atomic_load(loss_rows + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (L2-served)
    acc = wave_sum(acc);
    if ((c & 63) == 0) red[c >> 6] = acc;
    __syncthreads();
    if (c == 0) loss[0] = loss_scale * ((red[0] + red[1]) + (red[2] + red[3]));
}

static int plan(int B, int K, int nwg_hint, int* nbb, int* nwg, int* tpw) {
    *nbb = (B + QB - 1) / QB;
    const int tiles = K / QTILE;
    int want = nwg_hint > 0 ? nwg_hint : (128 / *nbb > 0 ? 128 / *nbb : 1);  // ~256 workgroups over both modalities
    if (want > tiles) want = tiles;
    *tpw = (tiles + want - 1) / want;
    *nwg = (tiles + *tpw - 1) / *tpw;
    return 0;
}

template <int NPL>
static int launch_qnce(QnceParams& p, int nbb, hipStream_t stream) {
    constexpr size_t lds = (size_t)NPL * (A1_SLOTS + A2_SLOTS) * sizeof(uint4) ;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)queue_nce_kernel<NPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_queue_nce_f32: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((queue_nce_kernel<NPL>), dim3(p.nwg, nbb, 2), dim3(256), lds, stream, p);
    return check_launch("trid_queue_nce_f32");
}

template <bool HASH>
static int launch_qnce_f16(QnceParams& p, int nbb, hipStream_t stream) {
    constexpr size_t lds = (size_t)2 * F_BUF * sizeof(uint4) + (HASH ? (F_HASH_SLOTS + 2) * sizeof(unsigned long long) : 0);
    static_assert((size_t)2 * F_BUF * sizeof(uint4) >= (size_t)4 * 32 * F_OPAD * sizeof(float), "the epilogue transpose must fit in the tile buffers");
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)queue_nce_f16_kernel<HASH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_queue_nce_f32: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL(queue_nce_f16_kernel<HASH>, dim3(p.nwg, nbb, 2), dim3(256), lds, stream, p);
    return check_launch("trid_queue_nce_f32");
}

}  // namespace trid

using namespace trid;

extern "C" long long trid_queue_nce_ws_floats(int B, int K, int C, int nwg_hint) {
    if (B <= 0 || B > QMAXB || K <= 0 || C != QC || K % QTILE != 0) return 0;
    int nbb, nwg, tpw;
    plan(B, K, nwg_hint, &nbb, &nwg, &tpw);
    return 2LL * nwg * (nbb * QB) * (QC + 1) + (K + 3) / 4;  // partials + the [K]-byte flag buffer of the large-batch path
}

extern "C" int trid_queue_nce_f32(const float* v_q, const float* t_q, const float* v_key, const float* t_key,
                                  const float* t_queue, const float* v_queue, const int64_t* id_queue, const int64_t* ids,
                                  float* loss_rows, float* dq, int B, int K, int C, float invT, float logit_bound,
                                  float gscale, int precision, int nwg_hint, float* ws, unsigned int* ticket, float* loss,
                                  float loss_scale, void* stream) {
    TRID_REQUIRE(v_q && t_q && v_key && t_key && t_queue && v_queue && id_queue && ids && loss_rows && dq && ws,
                 "trid_queue_nce_f32: null pointer");
    TRID_REQUIRE(B > 0 && K > 0 && invT > 0.f && logit_bound > 0.f, "trid_queue_nce_f32: bad sizes / scalars");
    TRID_REQUIRE((ticket == nullptr) == (loss == nullptr), "trid_queue_nce_f32: ticket and loss both or none");  // (before ANY launch)
    if (C != QC || K % QTILE != 0 || B > QMAXB) {
        set_error("trid_queue_nce_f32: built for C = %d, K %% %d == 0, B <= %d (got C = %d, K = %d, B = %d)", QC, QTILE, QMAXB, C, K, B);
        return TRID_E_UNSUPPORTED;
    }
    TRID_REQUIRE(aligned16(v_q) && aligned16(t_q) && aligned16(t_queue) && aligned16(v_queue) && aligned16(ws),
                 "trid_queue_nce_f32: queries / queues / workspace must be 16-byte aligned");
    QnceParams p;
    int nbb;
    plan(B, K, nwg_hint, &nbb, &p.nwg, &p.tiles_per_wg);
    p.q[0] = v_q;  p.queue[0] = t_queue;   // head.py:160-164: image queries against the text queue
    p.q[1] = t_q;  p.queue[1] = v_queue;   // head.py:166-170: text queries against the image queue
    p.B = B;
    p.Bp = nbb * QB;
    p.K = K;
    p.part_l = ws;
    p.part_o = ws + 2LL * p.nwg * p.Bp;
    p.id_queue = reinterpret_cast<const long long*>(id_queue);
    p.ids = reinterpret_cast<const long long*>(ids);
    p.hit = nullptr;
    const bool hashed = precision != 1 && precision != 3 && B <= F_HASH_MAXB;  // the fp16 kernel filters in-kernel
    if (!hashed) {  // batch-wide negative filter (head.py:148-157) as one byte per queue row, consumed 32 rows per tile
        uint8_t* flags = reinterpret_cast<uint8_t*>(ws + 2LL * p.nwg * p.Bp * (QC + 1));
        const int rc0 = trid_queue_hit_mask(id_queue, ids, flags, K, B, stream);
        if (rc0 != TRID_OK) return rc0;
        p.hit = flags;
    }
    const float log2e = 1.4426950408889634f;
    const float shift = logit_bound * invT;
    p.c1 = invT * log2e;
    p.c0 = -shift * log2e;
    const hipStream_t st = (hipStream_t)stream;
    const int rc = (precision == 1) ? launch_qnce<1>(p, nbb, st) : (precision == 3) ? launch_qnce<3>(p, nbb, st) : hashed ? launch_qnce_f16<true>(p, nbb, st) : launch_qnce_f16<false>(p, nbb, st);
    if (rc != TRID_OK) return rc;
    static_assert(QC == 256, "queue_nce_finish_kernel folds exactly four waves (red[0..3] / red[4..7]): one thread per channel of C = 256");
    hipLaunchKernelGGL(queue_nce_finish_kernel, dim3(B, 2), dim3(QC), 0, st, p, t_key, v_key, loss_rows, dq, invT, shift,
                       gscale * invT / (float)B, ticket, loss, loss_scale);
    return check_launch("trid_queue_nce_f32(finish)");
}
