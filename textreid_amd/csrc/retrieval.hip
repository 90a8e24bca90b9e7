// Retrieval match (evaluation.py:117-120) + top-k half of rank() (evaluation.py:14-19):
// similarity of Q queries against a gallery shard with the per-query top-k kept on
// device.  v1 structure: the gallery is walked in column chunks; each chunk's
// [Q, Gc] similarity tile is produced by the fp32 MFMA GEMM and immediately folded
// into the running per-row top-k (one wave per query row, register insertion sort +
// shuffle tournament), so only Q*k (value, index) pairs ever leave the device.

#include "common.h"

namespace trid {

constexpr int TOPK_MAX = 16;

template <int KK>
__global__ __launch_bounds__(256) void topk_merge_rows_kernel(const float* __restrict__ sim, int ld, int Q, int Gc,
                                                              long long col_offset, float* __restrict__ best_val,
                                                              long long* __restrict__ best_idx, int first) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    float v[KK];
    long long id[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) { v[i] = -INFINITY; id[i] = -1; }
    auto insert = [&](float x, long long xi) {
        // keep v sorted descending; ties -> lower index first
        if (x > v[KK - 1] || (x == v[KK - 1] && xi >= 0 && (id[KK - 1] < 0 || xi < id[KK - 1]))) {
            v[KK - 1] = x; id[KK - 1] = xi;
#pragma unroll
            for (int i = KK - 1; i > 0; --i) {
                const bool sw = v[i] > v[i - 1] || (v[i] == v[i - 1] && id[i] >= 0 && (id[i - 1] < 0 || id[i] < id[i - 1]));
                if (sw) {
                    const float tv = v[i]; v[i] = v[i - 1]; v[i - 1] = tv;
                    const long long ti = id[i]; id[i] = id[i - 1]; id[i - 1] = ti;
                }
            }
        }
    };
    const float* r = sim + (long long)row * ld;
    for (int j = lane; j < Gc; j += 64) insert(r[j], col_offset + j);
    if (!first && lane < KK) insert(best_val[(long long)row * KK + lane], best_idx[(long long)row * KK + lane]);
    // tournament: KK rounds, each picks the wave-wide best head and pops it
    for (int round = 0; round < KK; ++round) {
        float bv = v[0];
        long long bi = id[0];
        int bl = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const long long oi = __shfl_xor(bi, o, 64);
            const int ol = __shfl_xor(bl, o, 64);
            const bool take = ov > bv || (ov == bv && ((oi >= 0 && (bi < 0 || oi < bi)) || (oi == bi && ol < bl)));
            if (take) { bv = ov; bi = oi; bl = ol; }
        }
        if (lane == 0) {
            best_val[(long long)row * KK + round] = bv;
            best_idx[(long long)row * KK + round] = bi;
        }
        if (lane == bl) {
#pragma unroll
            for (int i = 0; i < KK - 1; ++i) { v[i] = v[i + 1]; id[i] = id[i + 1]; }
            v[KK - 1] = -INFINITY; id[KK - 1] = -1;
        }
    }
}

}  // namespace trid

using namespace trid;

static int topk_chunk_cols(int G) { return G < 8192 ? ((G + 3) / 4 * 4) : 8192; }

extern "C" long long trid_topk_ws_floats(int Q, int G, int k) {
    (void)k;
    return (long long)Q * topk_chunk_cols(G);
}

extern "C" int trid_sim_topk_f32(const float* q, const float* g, float* out_val, int64_t* out_idx, int Q, int G, int C,
                                 int k, long long idx_offset, float* ws, void* stream) {
    TRID_REQUIRE(q && g && out_val && out_idx && ws, "trid_sim_topk_f32: null pointer");
    TRID_REQUIRE(Q > 0 && G > 0 && C > 0 && C % 4 == 0, "trid_sim_topk_f32: bad shape (C%%4)");
    TRID_REQUIRE(k >= 1 && k <= TOPK_MAX && k <= G, "trid_sim_topk_f32: k must be in [1,%d] and <= G", TOPK_MAX);
    const int Gc = topk_chunk_cols(G);
    for (int c0 = 0; c0 < G; c0 += Gc) {
        const int n = (G - c0) < Gc ? (G - c0) : Gc;
        trid_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.A = q; d.B = g + (long long)c0 * C; d.C = ws;
        d.M = Q; d.N = n; d.K = C;
        d.lda = C; d.ldb = C; d.ldc = Gc;
        d.batch = 1; d.splits = 1; d.alpha = 1.f;
        d.a_mode = TRID_A_KC; d.b_mode = TRID_B_KC;
        int rc = trid_gemm_f32(&d, stream);
        if (rc) return rc;
        const dim3 grid((Q + 3) / 4), block(256);
        const int first = c0 == 0;
        const long long off = idx_offset + c0;
#define TRID_TOPK_CASE(KK)                                                                                          \
    case KK:                                                                                                        \
        hipLaunchKernelGGL(topk_merge_rows_kernel<KK>, grid, block, 0, (hipStream_t)stream, ws, Gc, Q, n, off, out_val, \
                           (long long*)out_idx, first);                                                             \
        break;
        switch (k) {
            TRID_TOPK_CASE(1) TRID_TOPK_CASE(2) TRID_TOPK_CASE(3) TRID_TOPK_CASE(4) TRID_TOPK_CASE(5) TRID_TOPK_CASE(6)
            TRID_TOPK_CASE(7) TRID_TOPK_CASE(8) TRID_TOPK_CASE(9) TRID_TOPK_CASE(10) TRID_TOPK_CASE(11) TRID_TOPK_CASE(12)
            TRID_TOPK_CASE(13) TRID_TOPK_CASE(14) TRID_TOPK_CASE(15) TRID_TOPK_CASE(16)
        }
#undef TRID_TOPK_CASE
        rc = check_launch("trid_sim_topk_f32");
        if (rc) return rc;
    }
    return TRID_OK;
}
