// fp32 MFMA GEMM / implicit-GEMM convolution family for gfx950 (CDNA4).
//
// One register-staged, LDS-tiled kernel template covers every dense contraction
// on the encode-and-match path:
//   C[m,n] (+)= alpha * sum_k A(m,k) * B(k,n)  (+ bias[n])
// with operand "loader modes" instead of CUDA-style transposed copies:
//   A_KC   A[m*lda + k]            rows K-contiguous       (activations, NT / NN)
//   A_MC   A[k*lda + m]            rows M-contiguous       (dY in weight-gradient, TN)
//   A_CONV NHWC image gather, k=(tap,c), 3x3 stride 1 pad 1 (implicit-GEMM fwd / dgrad)
//   B_KC   B[n*ldb + k]            weights [N,K]            (NT)
//   B_NC   B[k*ldb + n]            rows N-contiguous        (NN / TN)
//   B_CONV NHWC image gather on rows k=pixel, n=(tap,c)     (implicit-GEMM wgrad)
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Both LDS
// tiles are stored k-major ([k][m], [k][n]) so every MFMA operand fetch is a
// conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
// Epilogues: alpha, bias, accumulate, split-K slabs, and per-column
// (count-weighted mean, M2) partials for train-mode BatchNorm statistics.
//
// Reference ops served (file:line in /root/reference): every nn.Conv2d /
// nn.Linear / matmul on the hot path -- m_resnet.py:18-27,41-47,161-170 (convs),
// :114-133 (attention-pool projections), gru.py:36-43 (GRU projections),
// head.py:50-51,159-170 (embed layers, queue logits), losses.py:52-53,109-112.

#include <stdlib.h>

#include <mutex>

#include "gemm_common.h"

namespace trid {


template <int AMODE, int BMODE, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_kernel(GemmParams p) {
    constexpr int NTHREADS = WAVES_M * WAVES_N * 64;
    static_assert(NTHREADS == 256 || NTHREADS == 512, "4 or 8 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
    constexpr int LDA = BM + (AMODE == A_MC ? 4 : 1);
    constexpr int LDB = BN + (BMODE == B_KC ? 1 : 4);
    constexpr int NA = BM * BK / 4 / NTHREADS;  // float4 per thread per tile
    constexpr int NB = BN * BK / 4 / NTHREADS;
    static_assert(NA >= 1 && NB >= 1, "tile too small for 256 threads");

    // two LDS stages (A tile + B tile each): one barrier per K-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = BK * LDA + BK * LDB;

    if (p.gate != nullptr && *p.gate == 0) return;  // predicated launch (retrieval overflow fallback)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile order: n fastest so tiles sharing an A panel sit on one XCD's L2.
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

    // ---- per-thread loader state -------------------------------------------------
    // K-contiguous A: thread owns k-quad kq and rows ra[i];  M-contiguous: m-quad, k rows.
    constexpr int A_QPR = (AMODE == A_MC) ? BM / 4 : BK / 4;  // quads per tile row
    constexpr int A_RPP = NTHREADS / A_QPR;                   // rows per pass
    const int a_q = tid % A_QPR, a_r = tid / A_QPR;
    constexpr int B_QPR = (BMODE == B_KC) ? BK / 4 : BN / 4;
    constexpr int B_RPP = NTHREADS / B_QPR;
    const int b_q = tid % B_QPR, b_r = tid / B_QPR;

    // A_CONV: per-row pixel coordinates
    int a_y[NA], a_x[NA];
    if (AMODE == A_CONV) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            int m = m0 + a_r + i * A_RPP;
            uint32_t q = fdiv((uint32_t)m, p.fdW);
            a_x[i] = m - (int)q * p.W;
            uint32_t b = fdiv(q, p.fdH);
            a_y[i] = (int)q - (int)b * p.H;
        }
    }
    // B_CONV: per-thread fixed column quad -> (tap, c)
    int b_dy = 0, b_dx = 0, b_c = 0;
    bool b_colok = true;
    if (BMODE == B_CONV) {
        int j = n0 + 4 * b_q;
        b_colok = j < p.N;
        uint32_t tap = fdiv((uint32_t)j, p.fdC);
        b_c = j - (int)tap * p.Cin;
        b_dy = (int)tap / 3 - 1;
        b_dx = (int)tap % 3 - 1;
    }

    float4 ra[NA], rb[NB];

    auto load_tiles = [&](int k0) {
        // ---- A ----
        if (AMODE == A_KC) {
            const int k = k0 + 4 * a_q;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = m0 + a_r + i * A_RPP;
                if (m < p.M && k < k_end)
                    ra[i] = *reinterpret_cast<const float4*>(A + (long long)m * p.lda + k);
                else
                    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (AMODE == A_CONV) {
            const int k = k0 + 4 * a_q;
            const uint32_t tap = fdiv((uint32_t)k, p.fdC);
            const int c = k - (int)tap * p.Cin;
            const int dy = (int)tap / 3 - 1, dx = (int)tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = m0 + a_r + i * A_RPP;
                const int yy = a_y[i] + dy, xx = a_x[i] + dx;
                if (m < p.M && k < k_end && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                    ra[i] = *reinterpret_cast<const float4*>(A + (long long)(m + dy * p.W + dx) * p.Cin + c);
                else
                    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {  // A_MC
            const int m = m0 + 4 * a_q;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int k = k0 + a_r + i * A_RPP;
                if (m < p.M && k < k_end)
                    ra[i] = *reinterpret_cast<const float4*>(A + (long long)k * p.lda + m);
                else
                    ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // ---- B ----
        if (BMODE == B_KC) {
            const int k = k0 + 4 * b_q;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = n0 + b_r + i * B_RPP;
                if (n < p.N && k < k_end)
                    rb[i] = *reinterpret_cast<const float4*>(Bp + (long long)n * p.ldb + k);
                else
                    rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (BMODE == B_NC) {
            const int n = n0 + 4 * b_q;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k = k0 + b_r + i * B_RPP;
                if (n < p.N && k < k_end)
                    rb[i] = *reinterpret_cast<const float4*>(Bp + (long long)k * p.ldb + n);
                else
                    rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {  // B_CONV: row k is a pixel
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int k = k0 + b_r + i * B_RPP;
                uint32_t q = fdiv((uint32_t)k, p.fdW);
                const int x = k - (int)q * p.W;
                uint32_t b = fdiv(q, p.fdH);
                const int y = (int)q - (int)b * p.H;
                const int yy = y + b_dy, xx = x + b_dx;
                if (b_colok && k < k_end && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                    rb[i] = *reinterpret_cast<const float4*>(Bp + (long long)(k + b_dy * p.W + b_dx) * p.Cin + b_c);
                else
                    rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    auto store_tiles = [&](float* __restrict__ As, float* __restrict__ Bs) {
        if (AMODE == A_MC) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                *reinterpret_cast<float4*>(As + (a_r + i * A_RPP) * LDA + 4 * a_q) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = a_r + i * A_RPP;
                As[(4 * a_q + 0) * LDA + r] = ra[i].x;
                As[(4 * a_q + 1) * LDA + r] = ra[i].y;
                As[(4 * a_q + 2) * LDA + r] = ra[i].z;
                As[(4 * a_q + 3) * LDA + r] = ra[i].w;
            }
        }
        if (BMODE == B_KC) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = b_r + i * B_RPP;
                Bs[(4 * b_q + 0) * LDB + r] = rb[i].x;
                Bs[(4 * b_q + 1) * LDB + r] = rb[i].y;
                Bs[(4 * b_q + 2) * LDB + r] = rb[i].z;
                Bs[(4 * b_q + 3) * LDB + r] = rb[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<float4*>(Bs + (b_r + i * B_RPP) * LDB + 4 * b_q) = rb[i];
        }
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_off = wm * WM + (lane & 31);
    const int b_off = wn * WN + (lane & 31);
    const int khalf = lane >> 5;

    if (k_begin < k_end) {
        load_tiles(k_begin);
        store_tiles(smem, smem + BK * LDA);
        __syncthreads();
        int cur = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += BK, cur ^= 1) {
            const bool more = (k0 + BK) < k_end;
            if (more) load_tiles(k0 + BK);  // global loads in flight under the MFMAs
            const float* __restrict__ As = smem + cur * STAGE;
            const float* __restrict__ Bs = As + BK * LDA;
            // software-pipelined operand fetch: fragments of step kk+1 are read from LDS
            // before the MFMAs of step kk issue
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = As[khalf * LDA + a_off + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Bs[khalf * LDB + b_off + j * 32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int c = kk & 1, n = c ^ 1;
                if (kk + 1 < BK / 2) {
                    const int k = 2 * (kk + 1) + khalf;
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[n][i] = As[k * LDA + a_off + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[n][j] = Bs[k * LDB + b_off + j * 32];
                }
                // pin the order: hipcc otherwise sinks the prefetch reads next to their use
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
                if (kk == BK / 4 && more) {
                    // next tile -> the other LDS stage, in the shadow of the remaining MFMAs
                    float* nb = smem + (cur ^ 1) * STAGE;
                    store_tiles(nb, nb + BK * LDA);
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue ------------------------------------------------------------------
    const int row_base = m0 + wm * WM + 4 * khalf;
    const int col_base = n0 + wn * WN + (lane & 31);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col_base + j * 32;
        const float bv = (bias != nullptr && col < p.N) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float oldv[16];
            if (p.accumulate) {  // gather the old values first: 16 independent loads in flight
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
                if (p.res != nullptr && row < p.M && col < p.N) v += p.res[(long long)row * p.ldres + col];
                if (p.relu) v = fmaxf(v, 0.f);
                if (row < p.M && col < p.N) C[(long long)row * p.ldc + col] = v;
                acc[i][j][r] = v;
            }
        }
    }

    if (p.stats != nullptr) {
        // Per-column (mean, M2) over this tile's valid rows: two register passes,
        // half-wave exchange, then across the WAVES_M waves through LDS.
        __syncthreads();
        float* red = smem;  // [WAVES_M][BN]
        const int cnt = min(BM, p.M - m0);
        const float inv = 1.f / (float)cnt;
        float mean[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) s += acc[i][j][r];
                }
            s += __shfl_xor(s, 32, 64);
            if (khalf == 0) red[wm * BN + wn * WN + j * 32 + (lane & 31)] = s;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) s += red[w * BN + wn * WN + j * 32 + (lane & 31)];
            mean[j] = s * inv;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                    const float d = acc[i][j][r] - mean[j];
                    if (row < p.M) s += d * d;
                }
            s += __shfl_xor(s, 32, 64);
            if (khalf == 0) red[wm * BN + wn * WN + j * 32 + (lane & 31)] = s;
        }
        __syncthreads();
        if (wm == 0 && khalf == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = wn * WN + j * 32 + (lane & 31);
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES_M; ++w) s += red[w * BN + cl];
                const int col = n0 + cl;
                if (col < p.N) {
                    float* dst = p.stats + ((long long)mb * p.N + col) * 2;
                    dst[0] = mean[j];
                    dst[1] = s;
                }
            }
        }
    }
}

// ---- split-K slab reduction: C[i] = sum_s slab[s][i] (+ C[i]) ---------------------
// One workgroup = 64 float4 columns x 4 split lanes: the (possibly several hundred) slabs are
// walked by 4 waves in parallel with 4 independent loads in flight each, then folded in LDS
// in a fixed order (deterministic).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C,
                                                          long long n4, int splits, long long sSplit, int accumulate) {
    __shared__ float4 red[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 64; base < n4; base += (long long)gridDim.x * 64) {
        const long long i = base + cl;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        if (i < n4) {
            int k = sl;
            for (; k + 12 < splits; k += 16) {
                const float4 t0 = reinterpret_cast<const float4*>(slab + (long long)k * sSplit)[i];
                const float4 t1 = reinterpret_cast<const float4*>(slab + (long long)(k + 4) * sSplit)[i];
                const float4 t2 = reinterpret_cast<const float4*>(slab + (long long)(k + 8) * sSplit)[i];
                const float4 t3 = reinterpret_cast<const float4*>(slab + (long long)(k + 12) * sSplit)[i];
                a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
                a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
                a2.x += t2.x; a2.y += t2.y; a2.z += t2.z; a2.w += t2.w;
                a3.x += t3.x; a3.y += t3.y; a3.z += t3.z; a3.w += t3.w;
            }
            for (; k < splits; k += 4) {
                const float4 t0 = reinterpret_cast<const float4*>(slab + (long long)k * sSplit)[i];
                a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
            }
        }
        a0.x += a1.x + (a2.x + a3.x); a0.y += a1.y + (a2.y + a3.y);
        a0.z += a1.z + (a2.z + a3.z); a0.w += a1.w + (a2.w + a3.w);
        red[sl][cl] = a0;
        __syncthreads();
        if (sl == 0 && i < n4) {
            float4 s = red[0][cl];
            const float4 u = red[1][cl], v = red[2][cl], w = red[3][cl];
            s.x += u.x + (v.x + w.x); s.y += u.y + (v.y + w.y); s.z += u.z + (v.z + w.z); s.w += u.w + (v.w + w.w);
            if (accumulate) {
                const float4 t = reinterpret_cast<float4*>(C)[i];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            reinterpret_cast<float4*>(C)[i] = s;
        }
        __syncthreads();
    }
}

template <int AMODE, int BMODE, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch(GemmParams& p, hipStream_t stream) {
    p.mblocks = (p.M + BM - 1) / BM;
    p.nblocks = (p.N + BN - 1) / BN;
    dim3 grid((unsigned)(p.mblocks * p.nblocks), 1, (unsigned)(p.batch * p.splits));
    constexpr int LDA = BM + (AMODE == A_MC ? 4 : 1);
    constexpr int LDB = BN + (BMODE == B_KC ? 1 : 4);
    constexpr size_t lds = 2 * (size_t)(BK * LDA + BK * LDB) * sizeof(float);
    // once per kernel instantiation and process, safe under concurrent first calls from several host threads
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        if (lds > 48 * 1024)
            attr_err = hipFuncSetAttribute((const void*)gemm_kernel<AMODE, BMODE, BM, BN, WAVES_M, WAVES_N>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_gemm_f32: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((gemm_kernel<AMODE, BMODE, BM, BN, WAVES_M, WAVES_N>), grid, dim3(WAVES_M * WAVES_N * 64), lds, stream, p);
    return check_launch("trid_gemm_f32");
}

template <int AMODE, int BMODE>
static int dispatch_tile(GemmParams& p, hipStream_t stream) {
    // tile choice: fill 128-wide tiles when the dimension allows, shrink otherwise.
    // The BatchNorm-statistics epilogue always runs on 128-row tiles so that the
    // caller knows the partial count: ceil(M/128).
    int bm = p.M >= 96 ? 128 : (p.M > 32 ? 64 : 32);
    const int bn = p.N >= 96 ? 128 : (p.N > 32 ? 64 : 32);
    if (p.stats != nullptr) bm = 128;
    if (bm == 128) {
        if (bn == 128) {
            static const bool four = getenv("TRID_GEMM4") != nullptr;  // A/B switch: 4-wave 64x64 wave tiles
            if (!four) return launch<AMODE, BMODE, 128, 128, 2, 4>(p, stream);
            return launch<AMODE, BMODE, 128, 128, 2, 2>(p, stream);
        }
        if (bn == 64) return launch<AMODE, BMODE, 128, 64, 2, 2>(p, stream);
        return launch<AMODE, BMODE, 128, 32, 4, 1>(p, stream);
    }
    if (bn == 128) {
        if (bm == 64) return launch<AMODE, BMODE, 64, 128, 2, 2>(p, stream);
        return launch<AMODE, BMODE, 32, 128, 1, 4>(p, stream);
    }
    return launch<AMODE, BMODE, 64, 64, 2, 2>(p, stream);  // small x small, masked
}

}  // namespace trid

using namespace trid;

extern "C" int trid_gemm_f32(const trid_gemm_desc* d, void* stream) { return trid_gemm_launch(d, nullptr, nullptr, (hipStream_t)stream); }

int trid_gemm_launch(const trid_gemm_desc* d, const GemmFilter* filt, const int* gate, hipStream_t stream) {
    TRID_REQUIRE(d != nullptr, "trid_gemm_f32: null descriptor");
    TRID_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "trid_gemm_f32: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
    TRID_REQUIRE(d->A && d->B && (d->C || filt), "trid_gemm_f32: null operand");
    TRID_REQUIRE(aligned16(d->A) && aligned16(d->B) && aligned16(d->C), "trid_gemm_f32: operands must be 16-byte aligned");
    TRID_REQUIRE(d->a_mode >= 0 && d->a_mode <= 2 && d->b_mode >= 0 && d->b_mode <= 2, "trid_gemm_f32: bad loader mode");
    TRID_REQUIRE(d->batch >= 1 && d->splits >= 1, "trid_gemm_f32: batch/splits must be >= 1");
    TRID_REQUIRE(d->c_mask == nullptr, "trid_gemm_f32: c_mask is a trid_gemm_p16 epilogue");
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
    if (filt) p.filt = *filt;
    p.gate = gate;
    // contiguous-direction alignment (float4 loads)
    if (d->a_mode == A_KC) TRID_REQUIRE(d->K % 4 == 0 && d->lda % 4 == 0, "A_KC needs K%%4==0 and lda%%4==0 (K=%d lda=%lld)", d->K, d->lda);
    if (d->a_mode == A_MC) TRID_REQUIRE(d->M % 4 == 0 && d->lda % 4 == 0, "A_MC needs M%%4==0 and lda%%4==0 (M=%d)", d->M);
    if (d->b_mode == B_KC) TRID_REQUIRE(d->K % 4 == 0 && d->ldb % 4 == 0, "B_KC needs K%%4==0 and ldb%%4==0 (K=%d ldb=%lld)", d->K, d->ldb);
    if (d->b_mode == B_NC) TRID_REQUIRE(d->N % 4 == 0 && d->ldb % 4 == 0, "B_NC needs N%%4==0 and ldb%%4==0 (N=%d)", d->N);
    if (d->a_mode == A_CONV || d->b_mode == B_CONV) {
        TRID_REQUIRE(d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cin % 4 == 0, "conv gather needs H,W>0 and Cin%%4==0");
        if (d->a_mode == A_CONV) TRID_REQUIRE(d->K == 9 * d->Cin && d->M % (d->H * d->W) == 0, "A_CONV: K must be 9*Cin and M a multiple of H*W");
        if (d->b_mode == B_CONV) TRID_REQUIRE(d->N == 9 * d->Cin && d->K % (d->H * d->W) == 0, "B_CONV: N must be 9*Cin and K a multiple of H*W");
        p.fdW = make_fastdiv((uint32_t)d->W);
        p.fdH = make_fastdiv((uint32_t)d->H);
        p.fdC = make_fastdiv((uint32_t)d->Cin);
    }
    TRID_REQUIRE(!(d->stats && (d->splits != 1 || d->batch != 1)), "stats epilogue needs splits==1, batch==1");
    TRID_REQUIRE(!(d->splits > 1 && (d->accumulate || d->bias || d->residual || d->relu)), "split-K writes raw slabs: no bias/accumulate/residual/relu");
    TRID_REQUIRE(!(d->residual && d->batch != 1), "residual epilogue needs batch==1");
    int kc = (d->K + d->splits - 1) / d->splits;
    kc = (kc + BK - 1) / BK * BK;
    p.k_chunk = kc;
    p.sSplit = d->strideSplit;

    int rc;
    const int am = d->a_mode, bm = d->b_mode;
    // the split kernels address operands with 31-bit byte offsets (buffer loads): < 2 GB per operand and batch
    const long long a_elems = am == A_KC ? (long long)(d->M + 256) * d->lda
                            : am == A_MC ? (long long)d->K * d->lda : (long long)(d->M + 256 + 2 * d->W + 2) * d->Cin;
    const long long b_elems = bm == B_KC ? (long long)(d->N + 128) * d->ldb : bm == B_NC ? (long long)d->K * d->ldb : 0;
    const bool small_enough = a_elems < (1ll << 29) && b_elems < (1ll << 29);
    p.a_amax = d->a_amax;
    p.b_amax = d->b_amax;
    // fp16-split arithmetic also takes 32..63-column outputs (stem convs) on its 128x64 tile: half the tile is idle, still
    // 2x the exact fp32-MFMA kernel
    const int min_n = d->precision == 16 ? 32 : 64;
    if ((d->precision == 1 || d->precision == 3 || d->precision == 6 || d->precision == 16) && d->K % 8 == 0 && d->K >= 32 && d->M >= 64 && d->N >= min_n &&
        (d->M >= 96 || d->N >= 96) && (am != A_CONV || d->Cin % 8 == 0) && small_enough) {
        rc = gemm_bf16_dispatch(p, am, bm, d->precision, stream);
        if (rc != TRID_E_UNSUPPORTED) return rc;
    }
    if (filt) return TRID_E_UNSUPPORTED;  // the filter epilogue exists in the split kernel only
    if (am == A_KC && bm == B_KC) rc = dispatch_tile<A_KC, B_KC>(p, stream);
    else if (am == A_CONV && bm == B_KC) rc = dispatch_tile<A_CONV, B_KC>(p, stream);
    else if (am == A_KC && bm == B_NC) rc = dispatch_tile<A_KC, B_NC>(p, stream);
    else if (am == A_MC && bm == B_NC) rc = dispatch_tile<A_MC, B_NC>(p, stream);
    else if (am == A_MC && bm == B_CONV) rc = dispatch_tile<A_MC, B_CONV>(p, stream);
    else {
        set_error("trid_gemm_f32: unsupported loader combination a=%d b=%d", am, bm);
        return TRID_E_UNSUPPORTED;
    }
    return rc;
}

extern "C" int trid_slab_reduce_f32(const float* slab, float* C, long long n, int splits, long long strideSplit,
                                    int accumulate, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TRID_REQUIRE(slab && C && n > 0 && splits >= 1, "trid_slab_reduce_f32: bad arguments");
    TRID_REQUIRE(n % 4 == 0 && strideSplit % 4 == 0 && aligned16(slab) && aligned16(C), "trid_slab_reduce_f32: needs 16-byte alignment and n%%4==0");
    const long long n4 = n / 4;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for(n4, 64, 4096)), dim3(256), 0, stream, slab, C, n4, splits, strideSplit, accumulate);
    return check_launch("trid_slab_reduce_f32");
}
