// One launch per time step of the bias-free BiGRU (gru.py:66-82; nn.GRU cell equations): the recurrent product
// h @ W_hh^T (forward) / dgh @ W_hh (backward), the gate math, the state update and the running max-over-time all
// in ONE kernel for both directions.  The step boundary stays a kernel boundary on purpose: every step ends in
// an all-to-all exchange of the hidden state between the workgroups that own slices of W_hh, and on this chip a
// dependent launch (~1.5 us) is cheaper than any in-launch grid barrier (4-7 us; MI355X_MICROARCH price list),
// so "persistent" would be the slower design.  What the fusion removes is the [B, 3H] round trip through HBM,
// half of the launches, and the 128-wide GEMM tile that left most of the chip idle on a 128 x 1536 x 512 product.
//
// Work split: a workgroup owns 16 hidden units of one direction (their r, z, n rows of W_hh = three 16-wide MFMA
// column tiles) and 32 batch rows; its 16 waves are (16-row tile) x (K slice), see gru_step_fwd_kernel.
//
// Arithmetic: fp32-class through the fp16 two-plane split of split_common.h (3 MFMA products per multiply-add).
// W_hh is split ONCE per pass into the exact LDS image of the B fragments (trid_gru_pack_whh_f16), so a step only
// copies its 96 KB slice; h lies in (-1, 1), so its scale is fixed (2^13) and the producing step stores it already
// split ((hi | lo << 16) words) next to the fp32 state; the gradient dgh has no a-priori bound: each backward step
// publishes max|dgh| (one float per workgroup) and the next one scales and splits its A fragments with it.

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "split_common.h"

namespace trid {

constexpr int GU = 16;           // hidden units per workgroup
constexpr int GMB = 32;          // batch rows per workgroup (two 16-row MFMA tiles)
constexpr int GKQ = 8;           // K splits per tile: 16 waves = 2 tiles x 8
constexpr float GH_SC = 8192.f;  // 2^13: scale of the hidden state planes
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + __expf(-x)); }

// (hi | lo << 16) of x (already scaled)
__device__ __forceinline__ unsigned pack_hl(float x) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}

// eight packed words -> the hi and lo A fragments
__device__ __forceinline__ void unpack_hl(const uint4& u, const uint4& v, f16x8& hi, f16x8& lo) {
    const unsigned w[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = (w[2 * q] & 0xffffu) | (w[2 * q + 1] << 16);
        l[q] = (w[2 * q] >> 16) | (w[2 * q + 1] & 0xffff0000u);
    }
    hi = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// ---- W_hh [2][3H][H] -> fragment images.  Forward image of (direction, unit block): 16-byte slots
// [plane][gate][kstep][lane], lane = kg * 16 + n holding W[gate * H + u0 + n][32 kstep + 8 kg .. + 7];
// backward image: [plane][kstep][lane] holding W[32 kstep + 8 kg .. + 7][u0 + n] (kstep over the 3H rows).
__global__ void gru_pack_whh_kernel(const float* __restrict__ w, const float* __restrict__ amax, uint4* __restrict__ img_f,
                                    uint4* __restrict__ img_b, int H) {
    const int KS = H / 32;
    const long long per_blk = 3LL * KS * 64;  // slots of one plane of one (direction, unit block)
    const long long total = 2LL * (H / GU) * per_blk;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float sc = f16_scale_of(*amax);
    const int lane = (int)(i % 64), n = lane & 15, kg = lane >> 4;
    const long long blk = i / per_blk;  // d * (H / GU) + ublock
    const int d = (int)(blk / (H / GU)), u0 = (int)(blk % (H / GU)) * GU;
    const int rem = (int)((i % per_blk) / 64);
    const float* wd = w + (long long)d * 3 * H * H;
    float v[8];
    {   // forward: rem = gate * KS + kstep
        const int g = rem / KS, ks = rem % KS;
        const float* src = wd + ((long long)g * H + u0 + n) * H + 32 * ks + 8 * kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e] * sc;
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) f16_split2(v[2 * q], v[2 * q + 1], h[q], l[q]);
        uint4* dst = img_f + blk * 2 * per_blk + (long long)rem * 64 + lane;
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[per_blk] = make_uint4(l[0], l[1], l[2], l[3]);
    }
    {   // backward: rem = kstep over the 3H rows
        const float* src = wd + ((long long)32 * rem + 8 * kg) * H + u0 + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)e * H] * sc;
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) f16_split2(v[2 * q], v[2 * q + 1], h[q], l[q]);
        uint4* dst = img_b + blk * 2 * per_blk + (long long)rem * 64 + lane;
        dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
        dst[per_blk] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

struct GruFwdParams {
    const uint4* wimg;        // forward image
    const float* w_amax;      // max|W_hh| (the scale the image was built with)
    const unsigned* hp_in;    // [2][Bp][H] packed state entering the step
    unsigned* hp_out;         // [2][Bp][H] packed state leaving it
    float* h;                 // [2][B][H] fp32 state, in place
    const float* gi;          // [B*L][6H] input projections
    const long long* lengths;
    float* gates;             // [2][.][B][4H] slice of this step (or null)
    float* hprev;             // [2][.][B][H] slice of this step (or null)
    float* maxv;              // [B][2H]
    int* argt;                // [B][2H]
    int s, Lmax, L, B, Bp, H;
    long long gates_ds, hprev_ds;
};

// Workgroup = 1024 threads = 16 waves: wave (mt, kq) multiplies the 16-row tile mt by the K slice kq (2 tiles x 8
// slices), so every load of a step - the image copy (6 slots per thread), each wave's A fragments, each owner thread's
// cell inputs - is in flight at once and the step costs ONE memory round trip, not one per K chunk; the K-slice
// partials meet in LDS (over the image, after a barrier) and the cell update runs one (batch row, unit) pair per
// thread.  32 rows per workgroup (not 64): a step is bound by the bytes ONE CU pulls across the fabric (its A rows +
// its 96 KB image, at ~65-100 GB/s per CU), so the batch is spread over twice the CUs.
constexpr int GTHREADS = 1024;
constexpr int GCH_F = 2;   // A fragments a wave keeps in flight (k-steps per chunk): forward, K = H
constexpr int GCH_B = 6;   // backward, K = 3H

template <int NB>  // slots per thread in flight
__device__ __forceinline__ void stage_image(uint4* __restrict__ dst, const uint4* __restrict__ src, int n, int tid) {
    for (int i0 = tid; i0 < n; i0 += NB * GTHREADS) {
        uint4 t[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) t[q] = src[min(i0 + q * GTHREADS, n - 1)];
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (i0 + q * GTHREADS < n) dst[i0 + q * GTHREADS] = t[q];
    }
}

__global__ __launch_bounds__(GTHREADS) void gru_step_fwd_kernel(GruFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) uint4 wsm[];  // [plane][gate][kstep][64]; later the K-slice partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ub = blockIdx.x, d = blockIdx.y, mb = blockIdx.z;
    const int H = p.H, KS = H / 32, KQ = (KS + GKQ - 1) / GKQ;
    const int per_plane = 3 * KS * 64;
    const int mt = wave / GKQ, kq = wave % GKQ;
    const int row0 = mb * GMB + mt * 16;
    const int m = lane & 15, kg = lane >> 4;
    const bool live = row0 < p.B;  // (rows up to Bp are allocated and zero)
    const int ks0 = kq * KQ, ks1 = min(KS, ks0 + KQ);
    const unsigned* arow = p.hp_in + ((long long)d * p.Bp + (live ? row0 + m : 0)) * H + 8 * kg;
    uint4 au[GCH_F], av[GCH_F];
#pragma unroll
    for (int q = 0; q < GCH_F; ++q) {
        const int ks = min(ks0 + q, KS - 1);
        au[q] = *reinterpret_cast<const uint4*>(arow + 32 * ks);
        av[q] = *reinterpret_cast<const uint4*>(arow + 32 * ks + 4);
    }
    // ---- this thread's (batch row, unit) pair of the cell update: its inputs do not depend on the product
    const int er = tid >> 4, j = ub * GU + (tid & 15);
    const bool owner = er < GMB;  // the first GMB * 16 threads own one pair each
    const int eb = mb * GMB + er, ebc = min(owner ? eb : 0, p.B - 1);
    const int t = d == 0 ? p.s : p.Lmax - 1 - p.s;
    const long long hidx = ((long long)d * p.B + ebc) * H + j;
    const float hp = p.h[hidx];
    const bool act = (long long)t < p.lengths[ebc];
    const float* gir = p.gi + ((long long)ebc * p.L + t) * (6 * H) + (long long)d * 3 * H;
    const float gr = gir[j], gz = gir[H + j], gn = gir[2 * H + j];
    const int mc = ebc * 2 * H + d * H + j;
    const float cur = p.maxv[mc];
    stage_image<6>(wsm, p.wimg + ((long long)d * (H / GU) + ub) * 2 * per_plane, 2 * per_plane, tid);
    v4f acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = (v4f){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (live) {
        for (int c0 = ks0; c0 < ks1; c0 += GCH_F) {
            if (c0 > ks0) {  // (only when a K slice exceeds GCH_F k-steps)
#pragma unroll
                for (int q = 0; q < GCH_F; ++q) {
                    const int ks = min(c0 + q, KS - 1);
                    au[q] = *reinterpret_cast<const uint4*>(arow + 32 * ks);
                    av[q] = *reinterpret_cast<const uint4*>(arow + 32 * ks + 4);
                }
            }
#pragma unroll
            for (int q = 0; q < GCH_F; ++q) {
                const int ks = c0 + q;
                if (ks < ks1) {
                    f16x8 ah, al;
                    unpack_hl(au[q], av[q], ah, al);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const f16x8 bh = __builtin_bit_cast(f16x8, wsm[(g * KS + ks) * 64 + lane]);
                        const f16x8 bl = __builtin_bit_cast(f16x8, wsm[per_plane + (g * KS + ks) * 64 + lane]);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[g], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- K-slice partials -> LDS [kq][gate][row][unit] (accumulator register r = row 16 mt + 4 kg + r, unit m)
    __syncthreads();  // every wave is done with the image
    float* part = reinterpret_cast<float*>(wsm);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((kq * 3 + g) * GMB + 16 * mt + 4 * kg + r) * GU + m] = acc[g][r];
    __syncthreads();
    if (!owner || eb >= p.B) return;
    // ---- gates, state, running max (gru.py:48-63 semantics as in gru_cell_fwd_kernel)
    const float inv = 1.f / (f16_scale_of(*p.w_amax) * GH_SC);
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float* pg = part + (g * GMB + er) * GU + (tid & 15);
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < GKQ; ++q) a += pg[q * 3 * GMB * GU];  // fixed order
        gh[g] = a * inv;
    }
    if (p.hprev != nullptr) p.hprev[(long long)d * p.hprev_ds + (long long)eb * H + j] = hp;
    float hnew = hp;
    if (act) {
        const float rg = sigmoid_(gr + gh[0]);
        const float zg = sigmoid_(gz + gh[1]);
        const float hn_lin = gh[2];
        const float ng = tanhf(fmaf(rg, hn_lin, gn));
        hnew = (1.f - zg) * ng + zg * hp;
        p.h[hidx] = hnew;
        if (p.gates != nullptr) {
            float* gs = p.gates + (long long)d * p.gates_ds + (long long)eb * (4 * H);
            gs[j] = rg; gs[H + j] = zg; gs[2 * H + j] = ng; gs[3 * H + j] = hn_lin;
        }
        if (d == 0 ? (hnew > cur) : (hnew >= cur)) {  // first index wins on ties: forward walks t upward, reverse downward
            p.maxv[mc] = hnew;
            p.argt[mc] = t;
        }
    }
    p.hp_out[((long long)d * p.Bp + eb) * H + j] = pack_hl(hnew * GH_SC);
}

struct GruBwdParams {
    const uint4* wimg;      // backward image
    const float* w_amax;
    const float* dgh_in;    // [2][.][B][3H] slice of step s+1 (null at the last time step: nothing to propagate yet)
    const float* amax_in;   // [gridDim workgroups] per-workgroup max|dgh_in| published by step s+1
    float* amax_out;        // [gridDim workgroups] receives this step's (no atomics: 2048 waves on one word cost 20 us)
    const float* dout;      // [B][2H]
    const int* argt;
    const float* gates;     // slices of this step
    const float* hprev;
    const long long* lengths;
    float* dh;              // [2][B][H] gradient carried to the previous step, in place
    float* dGi;             // [B*L][6H]
    float* dgh_out;         // [2][.][B][3H] slice of this step
    int s, Lmax, L, B, H;
    long long gates_ds, hprev_ds, dgh_ds;
};

__global__ __launch_bounds__(GTHREADS) void gru_step_bwd_kernel(GruBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) uint4 wsm[];  // [plane][kstep over 3H][64]; later the K-slice partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ub = blockIdx.x, d = blockIdx.y, mb = blockIdx.z;
    const int H = p.H, KS = 3 * H / 32, KQ = (KS + GKQ - 1) / GKQ;
    const int per_plane = KS * 64;
    const int mt = wave / GKQ, kq = wave % GKQ;
    const int row0 = mb * GMB + mt * 16;
    const int m = lane & 15, kg = lane >> 4;
    const bool live = row0 < p.B;
    const bool more = p.dgh_in != nullptr;  // (uniform) false at the first processed step: nothing to propagate yet
    const int ks0 = kq * KQ, ks1 = min(KS, ks0 + KQ);
    // rows past B read any valid row (their results are dropped)
    const float* arow = more ? p.dgh_in + (long long)d * p.dgh_ds + (long long)min(row0 + m, p.B - 1) * (3 * H) + 8 * kg : nullptr;
    float4 au[GCH_B], av[GCH_B];
    if (more) {
#pragma unroll
        for (int q = 0; q < GCH_B; ++q) {
            const int ks = min(ks0 + q, KS - 1);
            au[q] = *reinterpret_cast<const float4*>(arow + 32 * ks);
            av[q] = *reinterpret_cast<const float4*>(arow + 32 * ks + 4);
        }
    }
    // ---- this thread's (batch row, unit) pair of the cell backward: its inputs do not depend on the product
    const int er = tid >> 4, j = ub * GU + (tid & 15);
    const bool owner = er < GMB;  // the first GMB * 16 threads own one pair each
    const int eb = mb * GMB + er, ebc = min(owner ? eb : 0, p.B - 1);
    const int t = d == 0 ? p.s : p.Lmax - 1 - p.s;
    const long long hidx = ((long long)d * p.B + ebc) * H + j;
    const float dh0 = p.dh[hidx];
    const bool act = (long long)t < p.lengths[ebc];
    const int mc = ebc * 2 * H + d * H + j;
    const float dov = p.argt[mc] == t ? p.dout[mc] : 0.f;
    const float* gs = p.gates + (long long)d * p.gates_ds + (long long)ebc * (4 * H);
    const float rg = gs[j], zg = gs[H + j], ng = gs[2 * H + j], hl = gs[3 * H + j];
    const float hpv = p.hprev[(long long)d * p.hprev_ds + (long long)ebc * H + j];
    float g_rec = 0.f;  // (dgh_in @ W_hh)[eb][j]
    if (more) {
        stage_image<6>(wsm, p.wimg + ((long long)d * (H / GU) + ub) * 2 * per_plane, 2 * per_plane, tid);
        float am = 0.f;  // fold of the producers' maxima (every wave folds all of them: a few hundred floats)
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        for (int i = lane; i < nwg; i += 64) am = fmaxf(am, p.amax_in[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
        const float sa = f16_scale_of(am);
        v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (live) {
            for (int c0 = ks0; c0 < ks1; c0 += GCH_B) {
                if (c0 > ks0) {  // (only when a K slice exceeds GCH_B k-steps: H > 512)
#pragma unroll
                    for (int q = 0; q < GCH_B; ++q) {
                        const int ks = min(c0 + q, KS - 1);
                        au[q] = *reinterpret_cast<const float4*>(arow + 32 * ks);
                        av[q] = *reinterpret_cast<const float4*>(arow + 32 * ks + 4);
                    }
                }
#pragma unroll
                for (int q = 0; q < GCH_B; ++q) {
                    const int ks = c0 + q;
                    if (ks < ks1) {
                        const float4 u = au[q], v = av[q];
                        unsigned hq[4], lq[4];
                        f16_split2(u.x * sa, u.y * sa, hq[0], lq[0]);
                        f16_split2(u.z * sa, u.w * sa, hq[1], lq[1]);
                        f16_split2(v.x * sa, v.y * sa, hq[2], lq[2]);
                        f16_split2(v.z * sa, v.w * sa, hq[3], lq[3]);
                        const f16x8 ah = __builtin_bit_cast(f16x8, make_uint4(hq[0], hq[1], hq[2], hq[3]));
                        const f16x8 al = __builtin_bit_cast(f16x8, make_uint4(lq[0], lq[1], lq[2], lq[3]));
                        const f16x8 bh = __builtin_bit_cast(f16x8, wsm[ks * 64 + lane]);
                        const f16x8 bl = __builtin_bit_cast(f16x8, wsm[per_plane + ks * 64 + lane]);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();  // every wave is done with the image
        float* part = reinterpret_cast<float*>(wsm);  // [kq][row][unit]
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(kq * GMB + 16 * mt + 4 * kg + r) * GU + m] = acc[r];
        __syncthreads();
        const float* pg = part + (er % GMB) * GU + (tid & 15);
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < GKQ; ++q) a += pg[q * GMB * GU];  // fixed order
        g_rec = a / (f16_scale_of(*p.w_amax) * sa);
    }
    // ---- cell backward (as gru_cell_bwd_kernel), with dh = carried gradient + this step's recurrent product
    float amax = 0.f;
    if (owner && eb < p.B) {
        float dhv = dh0 + g_rec;
        float* dgir = p.dGi + ((long long)eb * p.L + t) * (6 * H) + (long long)d * 3 * H;
        float* dghr = p.dgh_out + (long long)d * p.dgh_ds + (long long)eb * (3 * H);
        if (!act) {
            dgir[j] = 0.f; dgir[H + j] = 0.f; dgir[2 * H + j] = 0.f;
            dghr[j] = 0.f; dghr[H + j] = 0.f; dghr[2 * H + j] = 0.f;
            p.dh[hidx] = dhv;
        } else {
            dhv += dov;
            const float dn_pre = dhv * (1.f - zg) * (1.f - ng * ng);
            const float dz_pre = dhv * (hpv - ng) * zg * (1.f - zg);
            const float dr_pre = dn_pre * hl * rg * (1.f - rg);
            const float dn_h = dn_pre * rg;
            dgir[j] = dr_pre; dgir[H + j] = dz_pre; dgir[2 * H + j] = dn_pre;
            dghr[j] = dr_pre; dghr[H + j] = dz_pre; dghr[2 * H + j] = dn_h;
            p.dh[hidx] = dhv * zg;
            amax = fmaxf(fabsf(dr_pre), fmaxf(fabsf(dz_pre), fabsf(dn_h)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    __syncthreads();  // (the partials in LDS have been read)
    float* wmax = reinterpret_cast<float*>(wsm);
    if (lane == 0) wmax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
        float a = wmax[0];
        for (int i = 1; i < GTHREADS / 64; ++i) a = fmaxf(a, wmax[i]);
        p.amax_out[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = a;
    }
}

static bool gru_step_shape_ok(int B, int H) { return B > 0 && H >= 32 && H % 32 == 0 && H <= 768; }

}  // namespace trid

using namespace trid;

extern "C" long long trid_gru_whh_image_bytes(int H) {
    if (!gru_step_shape_ok(1, H)) return 0;
    return 2LL * (H / GU) * 2 * (3LL * (H / 32) * 64) * (long long)sizeof(uint4);  // one image (forward or backward)
}

extern "C" int trid_gru_step_workgroups(int B, int H) {
    if (!gru_step_shape_ok(B, H)) return 0;
    return (H / GU) * 2 * ((B + GMB - 1) / GMB);
}

extern "C" int trid_gru_pack_whh_f16(const float* w_hh, const float* w_amax, void* img_fwd, void* img_bwd, int H, void* stream) {
    TRID_REQUIRE(w_hh && w_amax && img_fwd && img_bwd, "trid_gru_pack_whh_f16: null pointer");
    if (!gru_step_shape_ok(1, H)) {
        set_error("trid_gru_pack_whh_f16: built for H %% 32 == 0, 32 <= H <= 768 (got %d)", H);
        return TRID_E_UNSUPPORTED;
    }
    TRID_REQUIRE(aligned16(img_fwd) && aligned16(img_bwd), "trid_gru_pack_whh_f16: images must be 16-byte aligned");
    const long long total = 2LL * (H / GU) * 3 * (H / 32) * 64;
    hipLaunchKernelGGL(gru_pack_whh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_hh, w_amax,
                       (uint4*)img_fwd, (uint4*)img_bwd, H);
    return check_launch("trid_gru_pack_whh_f16");
}

static int gru_lds_attr(const void* fn, size_t lds, const char* what) {
    // (called under std::call_once by the two launchers below)
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("%s: cannot reserve %zu B of LDS: %s", what, lds, hipGetErrorString(e));
        return (int)e;
    }
    return TRID_OK;
}

extern "C" int trid_gru_step_fwd_f32(const void* img_fwd, const float* w_amax, const void* hp_in, void* hp_out, float* h,
                                     const float* gi, const int64_t* lengths, float* gates, float* hprev, float* maxv,
                                     int32_t* argt, int s, int Lmax, int L, int B, int Bp, int H, long long gates_dstride,
                                     long long hprev_dstride, void* stream) {
    TRID_REQUIRE(img_fwd && w_amax && hp_in && hp_out && h && gi && lengths && maxv && argt, "trid_gru_step_fwd_f32: null pointer");
    TRID_REQUIRE(s >= 0 && s < Lmax && Lmax <= L && B > 0 && Bp >= B && Bp % 16 == 0, "trid_gru_step_fwd_f32: bad step/shape");
    if (!gru_step_shape_ok(B, H)) {
        set_error("trid_gru_step_fwd_f32: built for H %% 32 == 0, 32 <= H <= 768 (got %d)", H);
        return TRID_E_UNSUPPORTED;
    }
    TRID_REQUIRE(aligned16(img_fwd) && aligned16(hp_in) && aligned16(hp_out), "trid_gru_step_fwd_f32: images / packed state must be 16-byte aligned");
    static std::once_flag once;
    static int attr_rc = TRID_OK;
    std::call_once(once, [] { attr_rc = gru_lds_attr((const void*)gru_step_fwd_kernel, 768 * 192, "trid_gru_step_fwd_f32"); });
    if (attr_rc != TRID_OK) return attr_rc;
    GruFwdParams p;
    p.wimg = (const uint4*)img_fwd; p.w_amax = w_amax; p.hp_in = (const unsigned*)hp_in; p.hp_out = (unsigned*)hp_out; p.h = h;
    p.gi = gi; p.lengths = (const long long*)lengths; p.gates = gates; p.hprev = hprev; p.maxv = maxv; p.argt = argt;
    p.s = s; p.Lmax = Lmax; p.L = L; p.B = B; p.Bp = Bp; p.H = H; p.gates_ds = gates_dstride; p.hprev_ds = hprev_dstride;
    const size_t lds = std::max<size_t>((size_t)H * 192, GKQ * 3 * GMB * GU * sizeof(float));
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(H / GU, 2, (B + GMB - 1) / GMB), dim3(GTHREADS), lds, (hipStream_t)stream, p);
    return check_launch("trid_gru_step_fwd_f32");
}

extern "C" int trid_gru_step_bwd_f32(const void* img_bwd, const float* w_amax, const float* dgh_in, const float* amax_in,
                                     float* amax_out, const float* dout, const int32_t* argt, const float* gates,
                                     const float* hprev, const int64_t* lengths, float* dh, float* dGi, float* dgh_out, int s,
                                     int Lmax, int L, int B, int H, long long gates_dstride, long long hprev_dstride,
                                     long long dgh_dstride, void* stream) {
    TRID_REQUIRE(img_bwd && w_amax && amax_out && dout && argt && gates && hprev && lengths && dh && dGi && dgh_out,
                 "trid_gru_step_bwd_f32: null pointer");
    TRID_REQUIRE((dgh_in == nullptr) == (amax_in == nullptr), "trid_gru_step_bwd_f32: dgh_in and amax_in come together");
    TRID_REQUIRE(s >= 0 && s < Lmax && Lmax <= L && B > 0, "trid_gru_step_bwd_f32: bad step/shape");
    if (!gru_step_shape_ok(B, H)) {
        set_error("trid_gru_step_bwd_f32: built for H %% 32 == 0, 32 <= H <= 768 (got %d)", H);
        return TRID_E_UNSUPPORTED;
    }
    TRID_REQUIRE(aligned16(img_bwd) && (dgh_in == nullptr || aligned16(dgh_in)), "trid_gru_step_bwd_f32: image / dgh must be 16-byte aligned");
    static std::once_flag once;
    static int attr_rc = TRID_OK;
    std::call_once(once, [] { attr_rc = gru_lds_attr((const void*)gru_step_bwd_kernel, 768 * 192, "trid_gru_step_bwd_f32"); });
    if (attr_rc != TRID_OK) return attr_rc;
    GruBwdParams p;
    p.wimg = (const uint4*)img_bwd; p.w_amax = w_amax; p.dgh_in = dgh_in; p.amax_in = amax_in; p.amax_out = amax_out;
    p.dout = dout; p.argt = argt; p.gates = gates; p.hprev = hprev; p.lengths = (const long long*)lengths; p.dh = dh;
    p.dGi = dGi; p.dgh_out = dgh_out; p.s = s; p.Lmax = Lmax; p.L = L; p.B = B; p.H = H;
    p.gates_ds = gates_dstride; p.hprev_ds = hprev_dstride; p.dgh_ds = dgh_dstride;
    const size_t lds = std::max<size_t>((size_t)H * 192, GKQ * GMB * GU * sizeof(float));
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(H / GU, 2, (B + GMB - 1) / GMB), dim3(GTHREADS), lds, (hipStream_t)stream, p);
    return check_launch("trid_gru_step_bwd_f32");
}
