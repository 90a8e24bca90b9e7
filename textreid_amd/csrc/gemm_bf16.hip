// Split-precision GEMM / implicit-GEMM: fp32 operands, fp32 accumulate, bf16 MFMA.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA.  Each fp32
// operand is split ON THE FLY (while being staged into LDS) into NPL bf16 planes
//     x = hi + mid (+ lo),  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// and a product a*b is evaluated as the sum of the significant plane products:
//     NPL = 3 ("bf16x6"): hh + hm + mh + hl + lh + mm   (dropped terms <= 2^-26 |ab|: fp32-class)
//     NPL = 2 ("bf16x3"): hh + hm + mh                  (dropped terms ~ 2^-17 |ab|)
// all accumulated in the MFMA's fp32 accumulator, small terms first.  Same loader modes,
// epilogues (alpha / bias / accumulate / split-K / BatchNorm partials) and C ABI as gemm.hip.
//
// Tile 128x128x32, 8 waves (2x4, 64x32 per wave), single LDS stage + register prefetch;
// 3 workgroups per CU at NPL=3.  LDS image: 16-byte slots holding 8 consecutive k of one
// row, slot = kgroup*(128+1) + row, one image per plane -> every MFMA operand fetch is one
// conflict-free ds_read_b128 per plane.

#include "gemm_common.h"

namespace trid {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int T_ROWS = 128;          // BM = BN
constexpr int R1 = T_ROWS + 1;       // padded slots per k-group
constexpr int PLANE = 4 * R1;        // slots per plane (BK/8 = 4 k-groups)
constexpr int NT = 512;

template <int NPL>
__device__ __forceinline__ void split_store(const float (&v)[8], uint4* __restrict__ dst) {
    // dst: plane 0 slot; planes are PLANE slots apart
    unsigned short h[NPL][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float r = v[j];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const __bf16 b = (__bf16)r;  // round-to-nearest-even
            h[pl][j] = __builtin_bit_cast(unsigned short, b);
            r -= (float)b;  // exact in fp32
        }
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        uint4 u;
        u.x = (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16);
        u.y = (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16);
        u.z = (unsigned)h[pl][4] | ((unsigned)h[pl][5] << 16);
        u.w = (unsigned)h[pl][6] | ((unsigned)h[pl][7] << 16);
        dst[pl * PLANE] = u;
    }
}

template <int AMODE, int BMODE, int NPL>
__global__ __launch_bounds__(NT) void gemm_bf16s_kernel(GemmParams p) {
    constexpr int BM = T_ROWS, BN = T_ROWS;
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    uint4* As = smem4;
    uint4* Bs = smem4 + NPL * PLANE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, wave tile 64 x 32
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
    const float* __restrict__ bias = p.bias ? p.bias + (long long)bz * p.sBias : nullptr;

    // loader lanes: K-contiguous operands -> (kg = tid&3, row = tid>>2); M/N-contiguous -> (row = tid&127, kg = tid>>7)
    constexpr bool A_K = (AMODE != A_MC);
    constexpr bool B_K = (BMODE == B_KC);
    const int a_kg = A_K ? (tid & 3) : (tid >> 7), a_row = A_K ? (tid >> 2) : (tid & 127);
    const int b_kg = B_K ? (tid & 3) : (tid >> 7), b_row = B_K ? (tid >> 2) : (tid & 127);

    int a_y = 0, a_x = 0;
    if (AMODE == A_CONV) {
        const int m = m0 + a_row;
        const uint32_t q = fdiv((uint32_t)m, p.fdW);
        a_x = m - (int)q * p.W;
        const uint32_t b = fdiv(q, p.fdH);
        a_y = (int)q - (int)b * p.H;
    }
    int b_dy = 0, b_dx = 0, b_c = 0;
    if (BMODE == B_CONV) {
        const int j = n0 + b_row;
        const uint32_t tap = fdiv((uint32_t)j, p.fdC);
        b_c = j - (int)tap * p.Cin;
        b_dy = (int)tap / 3 - 1;
        b_dx = (int)tap % 3 - 1;
    }

    float ra[8], rb[8];

    auto load_tiles = [&](int k0) {
        // ---- A ----
        if (AMODE == A_KC) {
            const int m = m0 + a_row, k = k0 + 8 * a_kg;
            if (m < p.M && k < k_end) {
                const float4 u = *reinterpret_cast<const float4*>(A + (long long)m * p.lda + k);
                const float4 v = *reinterpret_cast<const float4*>(A + (long long)m * p.lda + k + 4);
                ra[0] = u.x; ra[1] = u.y; ra[2] = u.z; ra[3] = u.w; ra[4] = v.x; ra[5] = v.y; ra[6] = v.z; ra[7] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[j] = 0.f;
            }
        } else if (AMODE == A_CONV) {
            const int m = m0 + a_row, k = k0 + 8 * a_kg;
            const uint32_t tap = fdiv((uint32_t)k, p.fdC);
            const int c = k - (int)tap * p.Cin;
            const int dy = (int)tap / 3 - 1, dx = (int)tap % 3 - 1;
            const int yy = a_y + dy, xx = a_x + dx;
            if (m < p.M && k < k_end && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
                const float* src = A + (long long)(m + dy * p.W + dx) * p.Cin + c;
                const float4 u = *reinterpret_cast<const float4*>(src);
                const float4 v = *reinterpret_cast<const float4*>(src + 4);
                ra[0] = u.x; ra[1] = u.y; ra[2] = u.z; ra[3] = u.w; ra[4] = v.x; ra[5] = v.y; ra[6] = v.z; ra[7] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[j] = 0.f;
            }
        } else {  // A_MC: A[k*lda + m], lanes along m
            const int m = m0 + a_row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 8 * a_kg + j;
                ra[j] = (m < p.M && k < k_end) ? A[(long long)k * p.lda + m] : 0.f;
            }
        }
        // ---- B ----
        if (BMODE == B_KC) {
            const int n = n0 + b_row, k = k0 + 8 * b_kg;
            if (n < p.N && k < k_end) {
                const float4 u = *reinterpret_cast<const float4*>(Bp + (long long)n * p.ldb + k);
                const float4 v = *reinterpret_cast<const float4*>(Bp + (long long)n * p.ldb + k + 4);
                rb[0] = u.x; rb[1] = u.y; rb[2] = u.z; rb[3] = u.w; rb[4] = v.x; rb[5] = v.y; rb[6] = v.z; rb[7] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) rb[j] = 0.f;
            }
        } else if (BMODE == B_NC) {
            const int n = n0 + b_row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 8 * b_kg + j;
                rb[j] = (n < p.N && k < k_end) ? Bp[(long long)k * p.ldb + n] : 0.f;
            }
        } else {  // B_CONV: row k is a pixel, column n = (tap, c)
            const int n = n0 + b_row;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 8 * b_kg + j;
                const uint32_t q = fdiv((uint32_t)k, p.fdW);
                const int x = k - (int)q * p.W;
                const uint32_t b = fdiv(q, p.fdH);
                const int y = (int)q - (int)b * p.H;
                const int yy = y + b_dy, xx = x + b_dx;
                rb[j] = (n < p.N && k < k_end && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                            ? Bp[(long long)(k + b_dy * p.W + b_dx) * p.Cin + b_c]
                            : 0.f;
            }
        }
    };

    auto store_tiles = [&]() {
        split_store<NPL>(ra, As + a_kg * R1 + a_row);
        split_store<NPL>(rb, Bs + b_kg * R1 + b_row);
    };

    v16f acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int a_slot = wm * 64 + (lane & 31);
    const int b_slot = wn * 32 + (lane & 31);

    if (k_begin < k_end) {
        load_tiles(k_begin);
        store_tiles();
        __syncthreads();
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            const bool more = (k0 + BK) < k_end;
            if (more) load_tiles(k0 + BK);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[NPL][2], b[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const int s = pl * PLANE + (2 * ks + khalf) * R1;
                    a[pl][0] = __builtin_bit_cast(bf16x8, As[s + a_slot]);
                    a[pl][1] = __builtin_bit_cast(bf16x8, As[s + a_slot + 32]);
                    b[pl] = __builtin_bit_cast(bf16x8, Bs[s + b_slot]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (NPL == 3) {  // smallest terms first
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[NPL - 1], acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NPL - 1][i], b[0], acc[i], 0, 0, 0);
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0], acc[i], 0, 0, 0);
                }
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    // ---- epilogue (same contract as gemm.hip) -----------------------------------------
    const int row_base = m0 + wm * 64 + 4 * khalf;
    const int col = n0 + wn * 32 + (lane & 31);
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
            float v = p.alpha * acc[i][r] + bv;
            if (p.accumulate) v += oldv[r];
            if (row < p.M && col < p.N) C[(long long)row * p.ldc + col] = v;
            acc[i][r] = v;
        }
    }

    if (p.stats != nullptr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem4);  // [2][BN]
        const int cnt = min(BM, p.M - m0);
        const float inv = 1.f / (float)cnt;
        const int cl = wn * 32 + (lane & 31);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.M) s += acc[i][r];
            }
        s += __shfl_xor(s, 32, 64);
        if (khalf == 0) red[wm * BN + cl] = s;
        __syncthreads();
        const float mean = (red[cl] + red[BN + cl]) * inv;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2);
                const float d = acc[i][r] - mean;
                if (row < p.M) s += d * d;
            }
        s += __shfl_xor(s, 32, 64);
        if (khalf == 0) red[wm * BN + cl] = s;
        __syncthreads();
        if (wm == 0 && khalf == 0 && col < p.N) {
            float* dst = p.stats + ((long long)mb * p.N + col) * 2;
            dst[0] = mean;
            dst[1] = red[cl] + red[BN + cl];
        }
    }
}

template <int AMODE, int BMODE, int NPL>
static int launch_bf16(GemmParams& p, hipStream_t stream) {
    p.mblocks = (p.M + T_ROWS - 1) / T_ROWS;
    p.nblocks = (p.N + T_ROWS - 1) / T_ROWS;
    dim3 grid((unsigned)(p.mblocks * p.nblocks), 1, (unsigned)(p.batch * p.splits));
    constexpr size_t lds = (size_t)2 * NPL * PLANE * sizeof(uint4);
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16s_kernel<AMODE, BMODE, NPL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("trid_gemm_f32(split): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_bf16s_kernel<AMODE, BMODE, NPL>), grid, dim3(NT), lds, stream, p);
    return check_launch("trid_gemm_f32(split)");
}

template <int NPL>
static int dispatch_modes(GemmParams& p, int am, int bm, hipStream_t stream) {
    if (am == A_KC && bm == B_KC) return launch_bf16<A_KC, B_KC, NPL>(p, stream);
    if (am == A_CONV && bm == B_KC) return launch_bf16<A_CONV, B_KC, NPL>(p, stream);
    if (am == A_KC && bm == B_NC) return launch_bf16<A_KC, B_NC, NPL>(p, stream);
    if (am == A_MC && bm == B_NC) return launch_bf16<A_MC, B_NC, NPL>(p, stream);
    if (am == A_MC && bm == B_CONV) return launch_bf16<A_MC, B_CONV, NPL>(p, stream);
    return TRID_E_UNSUPPORTED;
}

int gemm_bf16_dispatch(GemmParams& p, int a_mode, int b_mode, int precision, hipStream_t stream) {
    if (precision == 6) return dispatch_modes<3>(p, a_mode, b_mode, stream);
    return dispatch_modes<2>(p, a_mode, b_mode, stream);
}

}  // namespace trid
