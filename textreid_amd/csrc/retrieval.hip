// Retrieval match (evaluation.py:117-120) + top-k half of rank() (evaluation.py:14-19):
// similarity of Q queries against a gallery shard with the per-query top-k kept on
// device.  v1 structure: the gallery is walked in column chunks; each chunk's
// [Q, Gc] similarity tile is produced by the fp32 MFMA GEMM and immediately folded
// into the running per-row top-k (one wave per query row, register insertion sort +
// shuffle tournament), so only Q*k (value, index) pairs ever leave the device.

#include "gemm_common.h"

namespace trid {

constexpr int TOPK_MAX = 16;

// Running top-k of one similarity row, held by one wave.  The k best (value, index) pairs live one per
// lane (lane i = i-th best, sorted by value descending, ties lower index first); the k-th value is the
// wave-uniform admission threshold.  Survivors are inserted one at a time with a ballot-popcount
// position and a one-lane shuffle shift.
template <int KK>
struct WaveTopk {
    float lv, t;
    long long li, ti;
    int lane;
    __device__ __forceinline__ void init(int lane_, const float* __restrict__ val, const long long* __restrict__ idx) {
        lane = lane_;
        lv = -INFINITY;
        li = -1;
        if (val != nullptr && lane < KK) { lv = val[lane]; li = idx[lane]; }
        t = __shfl(lv, KK - 1, 64);
        ti = __shfl(li, KK - 1, 64);
    }
    __device__ __forceinline__ void offer(float cv, long long ci) {  // wave-uniform candidate
        if (!(cv > t || (cv == t && (ti < 0 || ci < ti)))) return;
        const bool ahead = lane < KK && (lv > cv || (lv == cv && li >= 0 && li < ci));
        const int pos = __popcll(__ballot(ahead));
        const float uv = __shfl_up(lv, 1, 64);
        const long long ui = __shfl_up(li, 1, 64);
        if (lane > pos) { lv = uv; li = ui; }
        if (lane == pos) { lv = cv; li = ci; }
        if (lane >= KK) { lv = -INFINITY; li = -1; }
        t = __shfl(lv, KK - 1, 64);
        ti = __shfl(li, KK - 1, 64);
    }
    // every lane's candidate (xc, ci) with ok set is offered, in lane order
    __device__ __forceinline__ void offer_lanes(bool ok, float xc, long long ci) {
        unsigned long long bal = __ballot(ok && xc >= t);
        while (bal) {
            const int src = __ffsll((long long)bal) - 1;
            bal &= bal - 1;
            offer(__shfl(xc, src, 64), __shfl(ci, src, 64));
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ val, long long* __restrict__ idx) const {
        if (lane < KK) { val[lane] = lv; idx[lane] = li; }
    }
};

// Dense row scan: the row is streamed with UNROLL independent float4 loads per lane in flight and tested
// against the threshold with one ballot per batch, so after the first few hundred elements the kernel is
// a pure HBM stream (~k ln(G/k) survivors per row for unordered data).  Exact for any input order
// (sorted-ascending rows degrade to one insertion per element, never to a wrong answer).
template <int KK>
__global__ __launch_bounds__(256) void topk_merge_rows_kernel(const float* __restrict__ sim, int ld, int Q, int Gc,
                                                              long long col_offset, float* __restrict__ best_val,
                                                              long long* __restrict__ best_idx, int first,
                                                              const int* __restrict__ gate) {
    if (gate != nullptr && *gate == 0) return;
    constexpr int UNROLL = 8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    WaveTopk<KK> L;
    L.init(lane, first ? nullptr : best_val + (long long)row * KK, best_idx + (long long)row * KK);

    const float* r = sim + (long long)row * ld;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(r) & 15) == 0);
    for (int base = 0; base < Gc; base += 256 * UNROLL) {
        float4 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int j = base + u * 256 + lane * 4;
            if (vec_ok && j + 3 < Gc) {
                x[u] = *reinterpret_cast<const float4*>(r + j);
            } else {
                x[u].x = j < Gc ? r[j] : -INFINITY;
                x[u].y = j + 1 < Gc ? r[j + 1] : -INFINITY;
                x[u].z = j + 2 < Gc ? r[j + 2] : -INFINITY;
                x[u].w = j + 3 < Gc ? r[j + 3] : -INFINITY;
            }
        }
        bool hit = false;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) hit = hit || x[u].x >= L.t || x[u].y >= L.t || x[u].z >= L.t || x[u].w >= L.t;
        if (__ballot(hit) == 0ull) continue;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int j0 = base + u * 256 + lane * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xc = c == 0 ? x[u].x : c == 1 ? x[u].y : c == 2 ? x[u].z : x[u].w;
                // padding lanes carry -inf and are excluded by index
                L.offer_lanes(j0 + c < Gc, xc, col_offset + j0 + c);
            }
        }
    }
    L.store(best_val + (long long)row * KK, best_idx + (long long)row * KK);
}

// Candidate lists written by the GEMM's admission-filter epilogue (value, column) -> running top-k.
// A no-op when any list overflowed (*overflow != 0): the caller's gated dense passes redo the work.
// The row's counter is zeroed behind the merge: the next gallery segment's filter pass appends to an empty list.
template <int KK>
__global__ __launch_bounds__(256) void topk_merge_cands_kernel(const float2* __restrict__ cand, int* __restrict__ cnt,
                                                               int cap, int Q, long long idx_offset,
                                                               float* __restrict__ best_val, long long* __restrict__ best_idx,
                                                               const int* __restrict__ overflow) {
    if (*overflow != 0) return;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Q) return;
    const int n = cnt[row];
    if (n == 0) return;
    WaveTopk<KK> L;
    L.init(lane, best_val + (long long)row * KK, best_idx + (long long)row * KK);
    const float2* c = cand + (long long)row * cap;
    for (int base = 0; base < n; base += 64) {
        const bool ok = base + lane < n;
        const float2 pr = ok ? c[base + lane] : make_float2(-INFINITY, 0.f);
        L.offer_lanes(ok, pr.x, idx_offset + __float_as_int(pr.y));
    }
    L.store(best_val + (long long)row * KK, best_idx + (long long)row * KK);
    if (lane == 0) cnt[row] = 0;
}

// The running top-k as it stood after step 1, put back before the dense passes redo every column >= Gc (gate: only when a
// candidate list overflowed; null: unconditionally).
__global__ __launch_bounds__(256) void topk_restore_kernel(float* __restrict__ val, long long* __restrict__ idx,
                                                           const float* __restrict__ sval, const long long* __restrict__ sidx,
                                                           long long n, const int* __restrict__ gate) {
    if (gate != nullptr && *gate == 0) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        val[i] = sval[i];
        idx[i] = sidx[i];
    }
}

// Full descending argsort of each row (evaluation.py:14): one workgroup per row,
// bitonic network over (value, index) pairs in LDS; ties -> lower index first.
__global__ __launch_bounds__(256) void argsort_rows_desc_kernel(const float* __restrict__ sim, int ld, int G, int P,
                                                                long long* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sv = reinterpret_cast<float*>(smem_raw);
    int* si = reinterpret_cast<int*>(sv + P);
    const long long row = blockIdx.x;
    const float* r = sim + row * ld;
    for (int i = threadIdx.x; i < P; i += 256) {
        sv[i] = i < G ? r[i] : -INFINITY;
        si[i] = i < G ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = sv[i], b = sv[l];
                    const int ia = si[i], ib = si[l];
                    // "a should come before b" in the final descending order
                    const bool a_first = a > b || (a == b && ia < ib);
                    const bool up = (i & k) == 0;  // this sub-sequence sorted in final order?
                    if (up ? !a_first : a_first) {
                        sv[i] = b; sv[l] = a;
                        si[i] = ib; si[l] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < G; i += 256) out_idx[row * G + i] = si[i];
}

// One wave per query: first-hit rank (CMC) and average precision over R ranked gallery ids.
__global__ void rank_metrics_kernel(const long long* __restrict__ indices, const int64_t* __restrict__ q_pids,
                                    const int64_t* __restrict__ g_pids, int Q, int R, int32_t* __restrict__ first_hit,
                                    float* __restrict__ ap) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= Q) return;
    const int64_t qp = q_pids[row];
    int carried = 0, first = 0x7fffffff;
    float acc = 0.f;
    for (int c = 0; c < R; c += 64) {
        const int i = c + lane;
        bool m = false;
        if (i < R) m = g_pids[indices[(long long)row * R + i]] == qp;
        const unsigned long long bal = __ballot(m);
        if (m) {
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            acc += (float)(carried + before + 1) / (float)(i + 1);
        }
        if (bal != 0ull && first == 0x7fffffff) first = c + __ffsll((long long)bal) - 1;
        carried += __popcll(bal);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        first_hit[row] = first;
        ap[row] = acc / (float)carried;  // NaN when the query has no relevant gallery item, as the reference
    }
}

// cmc[t] = 100 * mean(first_hit < topk[t])
__global__ __launch_bounds__(256) void cmc_kernel(const int32_t* __restrict__ first_hit, int Q,
                                                  const int64_t* __restrict__ topk, int ntopk, float* __restrict__ cmc) {
    __shared__ float red[4];
    for (int t = 0; t < ntopk; ++t) {
        const int kk = (int)topk[t];
        float s = 0.f;
        for (int i = threadIdx.x; i < Q; i += 256) s += first_hit[i] < kk ? 1.f : 0.f;
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) cmc[t] = 100.f * (red[0] + red[1] + red[2] + red[3]) / (float)Q;
    }
}

// k-reciprocal re-rank term (evaluation.py:40-65): out[i,j] = alpha * |A_i n B_j| / |A_i u B_j| (+ base[i,j])
// for the top-k neighbour index sets A_i = qnn[i,:], B_j = gnn[j,:] (k <= 8, indices unique per row).
__global__ __launch_bounds__(256) void jaccard_add_kernel(const long long* __restrict__ qnn,
                                                          const long long* __restrict__ gnn,
                                                          const float* __restrict__ base, long long ldb,
                                                          float* __restrict__ out, int Q, int G, int k, float alpha) {
    const int i = blockIdx.y;
    long long a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = t < k ? qnn[(long long)i * k + t] : -1 - t;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < G; j += gridDim.x * blockDim.x) {
        int inter = 0;
        for (int u = 0; u < k; ++u) {
            const long long b = gnn[(long long)j * k + u];
#pragma unroll
            for (int t = 0; t < 8; ++t) inter += (a[t] == b) ? 1 : 0;
        }
        float v = alpha * (float)inter / (float)(2 * k - inter);
        if (base != nullptr) v += base[(long long)i * ldb + j];
        out[(long long)i * G + j] = v;
    }
}

}  // namespace trid

using namespace trid;

static int topk_chunk_cols(int G) { return G < 8192 ? ((G + 3) / 4 * 4) : 8192; }

// workspace: [Q x Gc similarity panel | later reused as Q candidate lists of Gc/2 (value, column) pairs]
//            [Q list counters][1 overflow flag][copy of the running top-k after step 1: Q x k indices, Q x k values]
static long long topk_ws_snap_offset(int Q, int G) { return ((long long)Q * topk_chunk_cols(G) + Q + 4 + 1) / 2 * 2; }
extern "C" long long trid_topk_ws_floats(int Q, int G, int k) {
    return topk_ws_snap_offset(Q, G) + 3ll * Q * k + 2;
}

template <int KK>
static void launch_scan(const float* ws, int ld, int Q, int n, long long off, float* out_val, int64_t* out_idx, int first,
                        const int* gate, hipStream_t stream) {
    hipLaunchKernelGGL(topk_merge_rows_kernel<KK>, dim3((Q + 3) / 4), dim3(256), 0, stream, ws, ld, Q, n, off, out_val,
                       (long long*)out_idx, first, gate);
}
template <int KK>
static void launch_cands(const float* cand, int* cnt, int cap, int Q, long long off, float* out_val, int64_t* out_idx,
                         const int* overflow, hipStream_t stream) {
    hipLaunchKernelGGL(topk_merge_cands_kernel<KK>, dim3((Q + 3) / 4), dim3(256), 0, stream, (const float2*)cand, cnt, cap, Q,
                       off, out_val, (long long*)out_idx, overflow);
}
#define TRID_TOPK_SWITCH(k, CALL)                                                                                  \
    switch (k) {                                                                                                   \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;            \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;            \
        case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break;      \
        case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break;    \
    }

// Similarity + per-query top-k of Q queries against G gallery rows.
//   1. the first Gc columns: GEMM -> [Q, Gc] panel -> streaming row scan -> sorted lists, whose k-th
//      value is every row's admission threshold;
//   2. all remaining columns in ONE GEMM whose epilogue stores nothing but the elements that reach the
//      row's threshold (~k (G-Gc)/Gc per row for unordered data) into per-row candidate lists;
//   3. the lists are merged into the running top-k.
// If a list overflows (adversarially ordered gallery) step 3 does nothing and the gated dense passes of
// step 4 (panel GEMM + scan per chunk, each a no-op unless the overflow flag is set) redo columns >= Gc,
// so the result is exact for any input.  Shapes / precisions the split kernel does not cover run the
// dense passes unconditionally.
// q16 / g16 (may be null): the same operands pre-split (P16 [ceil32(Q)][256] with zero padding rows, P16 [G][256], packed with
// q_amax / g_amax): step 2 then runs on the streaming kernel with the queries resident in registers (gemm_stream.hip FUSE 4)
// mode 0: everything (the gated dense passes of step 4 are enqueued behind the fused pass: no host round trip);
// mode 1: steps 1-3 only - the CALLER reads the overflow flag (ws[trid_topk_ws_flag_offset]) and runs mode 2 when it is set
//         (one 4-byte read instead of ~250 gated no-op launches = 1.4 ms of a 17 ms match at Q = 1e4, G = 1e6);
// mode 2: the dense passes over the columns >= Gc, unconditionally (after a mode-1 call whose lists overflowed).
static int sim_topk(const float* q, const float* g, const void* q16, const void* g16, float* out_val, int64_t* out_idx, int Q, int G, int C,
                    int k, long long idx_offset, int precision, const float* q_amax, const float* g_amax, float* ws, int mode, hipStream_t stream) {
    TRID_REQUIRE(q && g && out_val && out_idx && ws, "trid_sim_topk_f32: null pointer");
    TRID_REQUIRE(Q > 0 && G > 0 && C > 0 && C % 4 == 0, "trid_sim_topk_f32: bad shape (C%%4)");
    TRID_REQUIRE(k >= 1 && k <= TOPK_MAX && k <= G, "trid_sim_topk_f32: k must be in [1,%d] and <= G", TOPK_MAX);
    if (precision == 16 && !(q_amax && g_amax)) precision = 6;  // the fp16 split needs the operands' magnitudes
    const int Gc = topk_chunk_cols(G);
    int* cnt = reinterpret_cast<int*>(ws + (long long)Q * Gc);
    int* overflow = cnt + Q;
    const int cap = Gc / 2;

    auto panel_gemm = [&](int c0, int n, const int* gate) {
        trid_gemm_desc d;
        memset(&d, 0, sizeof(d));
        d.A = q; d.B = g + (long long)c0 * C; d.C = ws;
        d.M = Q; d.N = n; d.K = C;
        d.lda = C; d.ldb = C; d.ldc = Gc;
        d.batch = 1; d.splits = 1; d.alpha = 1.f;
        d.a_mode = TRID_A_KC; d.b_mode = TRID_B_KC;
        d.precision = precision;
        d.a_amax = q_amax; d.b_amax = g_amax;
        return trid_gemm_launch(&d, nullptr, gate, stream);
    };
    auto dense_pass = [&](int c0, const int* gate) {
        const int n = (G - c0) < Gc ? (G - c0) : Gc;
        int rc = panel_gemm(c0, n, gate);
        if (rc) return rc;
        const int first = c0 == 0;
        const long long off = idx_offset + c0;
#define TRID_CALL(KK) launch_scan<KK>(ws, Gc, Q, n, off, out_val, out_idx, first, gate, stream)
        TRID_TOPK_SWITCH(k, TRID_CALL)
#undef TRID_CALL
        return check_launch("trid_sim_topk_f32");
    };

    int rc = TRID_OK;
    if (mode != 2) rc = dense_pass(0, nullptr);
    if (rc || G <= Gc) return rc;

    // pre-split operands: the columns >= Gc go through the streaming filter in SEGMENTS of growing length (x 8), each merged
    // before the next starts - a query's threshold is then the k-th best of everything before the segment, and it admits
    // ~k (s1 - s0) / s0 <= 7 k candidates per segment instead of ~k (G - Gc) / Gc in all (G = 1e6, k = 10: 150 per query
    // instead of 1210; the appends and their merge were 15 % of the pass)
    const bool presplit = q16 != nullptr && g16 != nullptr;
    long long* snap_idx = reinterpret_cast<long long*>(ws + topk_ws_snap_offset(Q, G));
    float* snap_val = reinterpret_cast<float*>(snap_idx + (long long)Q * k);
    const long long nk = (long long)Q * k;
    auto restore = [&](const int* gate) {
        hipLaunchKernelGGL(topk_restore_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, stream, out_val, (long long*)out_idx,
                           (const float*)snap_val, (const long long*)snap_idx, nk, gate);
        return check_launch("trid_sim_topk: restore");
    };

    bool fused = false;
    if (mode != 2) {
        hipError_t e = hipMemsetAsync(cnt, 0, (size_t)(Q + 1) * sizeof(int), stream);
        if (e != hipSuccess) { set_error("trid_sim_topk_f32: memset failed: %s", hipGetErrorString(e)); return (int)e; }
        GemmFilter f;
        f.thr = out_val + (k - 1); f.thr_stride = k;
        f.cnt = cnt; f.cand = ws; f.cap = cap; f.col0 = Gc; f.overflow = overflow;
        if (presplit) {
            // (the dense passes that redo the columns >= Gc after an overflow must start from the top-k of step 1)
            e = hipMemcpyAsync(snap_val, out_val, (size_t)nk * sizeof(float), hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(snap_idx, out_idx, (size_t)nk * sizeof(long long), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) { set_error("trid_sim_topk: copy failed: %s", hipGetErrorString(e)); return (int)e; }
            static const int seg_env = getenv("TRID_TOPK_SEGMENTS") ? atoi(getenv("TRID_TOPK_SEGMENTS")) : 1;  // (0: one segment)
            for (long long s0 = Gc; s0 < G && rc == TRID_OK;) {
                long long s1 = seg_env ? s0 * 8 : G;
                if (s1 >= G || G - s1 < s1 / 4) s1 = G;  // (no short last segment)
                f.col0 = (int)s0;
                rc = stream_topk_filter(reinterpret_cast<const char*>(g16) + (size_t)s0 * 1024, g_amax, q16, q_amax, (int)(s1 - s0), Q, f, stream);
                if (rc) break;
#define TRID_CALL(KK) launch_cands<KK>(ws, cnt, cap, Q, idx_offset, out_val, out_idx, overflow, stream)
                TRID_TOPK_SWITCH(k, TRID_CALL)
#undef TRID_CALL
                rc = check_launch("trid_sim_topk_f32");
                s0 = s1;
            }
            if (rc) return rc;
            fused = true;
        } else {
            trid_gemm_desc d;
            memset(&d, 0, sizeof(d));
            d.A = q; d.B = g + (long long)Gc * C; d.C = nullptr;
            d.M = Q; d.N = G - Gc; d.K = C;
            d.lda = C; d.ldb = C; d.ldc = 0;
            d.batch = 1; d.splits = 1; d.alpha = 1.f;
            d.a_mode = TRID_A_KC; d.b_mode = TRID_B_KC;
            d.precision = precision;
            d.a_amax = q_amax; d.b_amax = g_amax;
            rc = trid_gemm_launch(&d, &f, nullptr, stream);
            if (rc == TRID_OK) {
                fused = true;
#define TRID_CALL(KK) launch_cands<KK>(ws, cnt, cap, Q, idx_offset, out_val, out_idx, overflow, stream)
                TRID_TOPK_SWITCH(k, TRID_CALL)
#undef TRID_CALL
                rc = check_launch("trid_sim_topk_f32");
                if (rc) return rc;
            } else if (rc != TRID_E_UNSUPPORTED) {
                return rc;
            }
        }
    }
    // dense passes over the remaining chunks: unconditional without the fused path, gated on overflow with it
    if (mode == 1 && fused) return TRID_OK;
    if (mode == 1) {  // (the fused pass did not apply: the dense passes are the result - tell the caller not to redo them)
        hipError_t e = hipMemsetAsync(overflow, 0, sizeof(int), stream);
        if (e != hipSuccess) { set_error("trid_sim_topk: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    }
    if (presplit && (mode == 2 || fused)) {  // (merged segments before the one that overflowed are in the running top-k: back to step 1)
        rc = restore(mode == 2 ? nullptr : overflow);
        if (rc) return rc;
    }
    for (int c0 = Gc; c0 < G; c0 += Gc) {
        rc = dense_pass(c0, (fused && mode == 0) ? overflow : nullptr);
        if (rc) return rc;
    }
    return TRID_OK;
}

extern "C" long long trid_topk_ws_flag_offset(int Q, int G) { return (long long)Q * topk_chunk_cols(G) + Q; }

extern "C" int trid_sim_topk_f32(const float* q, const float* g, float* out_val, int64_t* out_idx, int Q, int G, int C,
                                 int k, long long idx_offset, int precision, const float* q_amax, const float* g_amax, float* ws,
                                 void* stream_) {
    return sim_topk(q, g, nullptr, nullptr, out_val, out_idx, Q, G, C, k, idx_offset, precision, q_amax, g_amax, ws, 0, (hipStream_t)stream_);
}

extern "C" int trid_sim_topk_p16(const float* q, const float* g, const void* q16, const void* g16, float* out_val, int64_t* out_idx, int Q, int G,
                                 int k, long long idx_offset, const float* q_amax, const float* g_amax, float* ws, int mode, void* stream_) {
    TRID_REQUIRE(mode >= 0 && mode <= 2, "trid_sim_topk_p16: mode must be 0, 1 or 2");
    // (mode 2 reads neither pre-split operand, but the call must say that the mode-1 pass before it ran on them: the segments it
    // merged before a list overflowed are undone from the copy in ws)
    TRID_REQUIRE(q16 && g16 && q_amax && g_amax && aligned16(q16) && aligned16(g16), "trid_sim_topk_p16: the pre-split operands and their amax scalars are needed");
    TRID_REQUIRE((long long)G * 1024 < (1ll << 31), "trid_sim_topk_p16: the gallery shard must stay below 2 GB (G <= 2097151 rows of 256)");
    return sim_topk(q, g, q16, g16, out_val, out_idx, Q, G, 256, k, idx_offset, 16, q_amax, g_amax, ws, mode, (hipStream_t)stream_);
}

extern "C" int trid_topk_rows_f32(const float* sim, int ld, int Q, int G, int k, float* out_val, int64_t* out_idx,
                                  void* stream) {
    TRID_REQUIRE(sim && out_val && out_idx && Q > 0 && G > 0 && ld >= G, "trid_topk_rows_f32: bad arguments");
    TRID_REQUIRE(k >= 1 && k <= TOPK_MAX && k <= G, "trid_topk_rows_f32: k must be in [1,%d] and <= G", TOPK_MAX);
#define TRID_CALL(KK) launch_scan<KK>(sim, ld, Q, G, 0LL, out_val, out_idx, 1, nullptr, (hipStream_t)stream)
    TRID_TOPK_SWITCH(k, TRID_CALL)
#undef TRID_CALL
    return check_launch("trid_topk_rows_f32");
}

namespace trid {
int argsort_rows_desc_large(const float* sim, int ld, int Q, int G, int64_t* out_idx, void* ws, long long ws_bytes, void* stream);
}

extern "C" int trid_argsort_rows_desc_f32(const float* sim, int ld, int Q, int G, int64_t* out_idx, void* ws, long long ws_bytes,
                                          void* stream) {
    TRID_REQUIRE(sim && out_idx && Q > 0 && G > 0 && ld >= G, "trid_argsort_rows_desc_f32: bad arguments");
    if (G > 16384) return trid::argsort_rows_desc_large(sim, ld, Q, G, out_idx, ws, ws_bytes, stream);  // argsort_large.hip
    int P = 2;
    while (P < G) P <<= 1;
    const size_t lds = (size_t)P * 8;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)argsort_rows_desc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("trid_argsort_rows_desc_f32: cannot reserve %zu B of LDS", lds); return (int)e; }
    }
    hipLaunchKernelGGL(argsort_rows_desc_kernel, dim3(Q), dim3(256), lds, (hipStream_t)stream, sim, ld, G, P,
                       (long long*)out_idx);
    return check_launch("trid_argsort_rows_desc_f32");
}

extern "C" int trid_rank_metrics(const int64_t* indices, const int64_t* q_pids, const int64_t* g_pids, int Q, int R,
                                 int32_t* first_hit, float* ap, const int64_t* topk, int ntopk, float* cmc, void* stream) {
    TRID_REQUIRE(indices && q_pids && g_pids && first_hit && ap && topk && cmc && Q > 0 && R > 0 && ntopk > 0,
                 "trid_rank_metrics: bad arguments");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3((Q + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const long long*)indices,
                       q_pids, g_pids, Q, R, first_hit, ap);
    hipLaunchKernelGGL(cmc_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, first_hit, Q, topk, ntopk, cmc);
    return check_launch("trid_rank_metrics");
}

extern "C" int trid_jaccard_add_f32(const int64_t* qnn, const int64_t* gnn, const float* base, long long ldb, float* out,
                                    int Q, int G, int k, float alpha, void* stream) {
    TRID_REQUIRE(qnn && gnn && out && Q > 0 && G > 0 && k >= 1 && k <= 8, "trid_jaccard_add_f32: bad arguments (k<=8)");
    TRID_REQUIRE(base == nullptr || ldb >= G, "trid_jaccard_add_f32: bad base stride");
    dim3 grid((G + 255) / 256 > 64 ? 64 : (G + 255) / 256, Q);
    hipLaunchKernelGGL(jaccard_add_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const long long*)qnn,
                       (const long long*)gnn, base, ldb, out, Q, G, k, alpha);
    return check_launch("trid_jaccard_add_f32");
}
