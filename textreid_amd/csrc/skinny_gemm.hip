// Skinny GEMMs: C[M][N] (+)= alpha * A[M][K] . B + bias for M <= 128 rows (M = the batch: the attention pool's q / c
// projections, m_resnet.py:114-133, the embedding layers, head.py:50-51,126-129, and their data gradients).
//
// With 128 rows a tiled GEMM has ONE row of output tiles: [128 x 2048] x [2048 x 2048] is 16 workgroups of a 128 x 128
// kernel on a 256-CU chip (104 us measured for 1 GFLOP).  The work here is weight streaming: every weight is used for
// 128 rows once.  So: a workgroup owns 32 output columns and up to 64 rows, its 8 waves split the REDUCTION eight ways
// (each wave runs v_mfma_f32_32x32x2_f32 - exact fp32 - over its K slice with both operands loaded straight from global
// memory into registers, 16 bytes per lane and load), the eight partial tiles meet in LDS and are folded in a fixed order.
// No LDS staging of operands, no barriers in the loop; the chip sees N/32 x ceil(M/64) workgroups of 8 waves.
//
// k order inside a wave: lane (row i, half h) loads k = 8 j + 4 h .. + 3 of its row as one float4, step t of group j feeds
// the MFMA k-pair (8 j + t, 8 j + 4 + t): a permutation of the reduction order, identical for A and B.

#include <algorithm>

#include "gemm_common.h"

namespace trid {

struct SkinnyParams {
    const float* A;   // [M][lda]
    const float* B;   // BMODE 0 (B_KC): [N][ldb], k contiguous;  BMODE 1 (B_NC): [K][ldb], n contiguous
    float* C;         // [M][ldc]
    const float* bias;
    int M, N, K;
    long long lda, ldb, ldc;
    float alpha;
    int accumulate;
    int kslice;       // k range of one wave (multiple of 8)
    long long sA, sB, sC, sBias;  // batch strides (elements); blockIdx.z = batch index
};

template <int BMODE>
__global__ __launch_bounds__(512) void skinny_gemm_f32_kernel(SkinnyParams p) {
    __shared__ float part[7][2][16][64];  // partial tiles of waves 1..7: [wave][row block][acc register][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 64;
    const int nrb = (p.M - m0 > 32) ? 2 : 1;  // row blocks of this workgroup that hold rows

    const long long z = blockIdx.z;
    const float* Ab = p.A + z * p.sA;
    const float* Bb = p.B + z * p.sB;
    float* Cb = p.C + z * p.sC;
    // (descriptor ranges: the last row ends after its K / N elements, not after a full pitch - batched operands interleave)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (unsigned)(((size_t)(p.M - 1) * p.lda + p.K) * 4), 0x00020000);
    const size_t b_bytes = BMODE == 0 ? ((size_t)(p.N - 1) * p.ldb + p.K) * 4 : ((size_t)(p.K - 1) * p.ldb + p.N) * 4;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (unsigned)b_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    const int k_begin = wave * p.kslice;
    const int k_end = min(p.K, k_begin + p.kslice);
    // per-lane row bases (bytes); a row / column beyond the matrix reads zeros
    unsigned a_row[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int m = m0 + rb * 32 + i;
        a_row[rb] = m < p.M ? (unsigned)((size_t)m * p.lda * 4) : OOB;
    }
    const int n = n0 + i;
    const unsigned b_row = BMODE == 0 ? (n < p.N ? (unsigned)((size_t)n * p.ldb * 4) : OOB) : (n < p.N ? (unsigned)(n * 4) : OOB);
    const unsigned ldb4 = (unsigned)(p.ldb * 4);

    v16f acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;

    typedef float f4 __attribute__((ext_vector_type(4)));
    struct Frag {
        f4 a[2], b;
    };
    auto load = [&](int k0, Frag& f) {
        const int k = k0 + 4 * h;
        const bool kin = k < k_end;  // (K % 4 == 0: a float4 lies inside the row or outside the slice)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
            f.a[rb] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (kin && rb < nrb) ? a_row[rb] + (unsigned)k * 4u : OOB, 0, 0));
        if (BMODE == 0) {
            f.b = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsB, kin ? b_row + (unsigned)k * 4u : OOB, 0, 0));
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                f.b[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, kin ? b_row + (unsigned)(k + t) * ldb4 : OOB, 0, 0));
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][t], f.b[t], acc[0], 0, 0, 0);
            if (nrb == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][t], f.b[t], acc[1], 0, 0, 0);  // (workgroup-uniform)
        }
    };
    // four 8-deep groups in flight: the loads of groups j+4 .. j+7 are issued before the MFMAs of groups j .. j+3
    Frag f0[4], f1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load(k_begin + 8 * u, f0[u]);
    for (int k0 = k_begin; k0 < k_end; k0 += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) load(k0 + 32 + 8 * u, f1[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) mma(f0[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) load(k0 + 64 + 8 * u, f0[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) mma(f1[u]);
    }
    if (wave > 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave - 1][rb][r][lane] = acc[rb][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (int w = 0; w < 7; ++w)  // (fixed order; one partial tile - 32 LDS reads - in flight at a time)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] += part[w][rb][r][lane];
    const float bv = (p.bias != nullptr && n < p.N) ? p.bias[z * p.sBias + n] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < p.M && n < p.N) {
                float* dst = Cb + (size_t)m * p.ldc + n;
                float v = p.alpha * acc[rb][r] + bv;
                if (p.accumulate) v += *dst;
                *dst = v;
            }
        }
}

}  // namespace trid

using namespace trid;

extern "C" int trid_skinny_gemm_f32(const float* A, long long lda, const float* B, long long ldb, int b_mode, float* C, long long ldc,
                                    const float* bias, int M, int N, int K, float alpha, int accumulate, int batch, long long strideA,
                                    long long strideB, long long strideC, long long strideBias, void* stream) {
    TRID_REQUIRE(A && B && C && M > 0 && M <= 128 && N > 0 && K > 0 && batch >= 1 && batch <= 65535, "trid_skinny_gemm_f32: needs 0 < M <= 128, 1 <= batch <= 65535 (M=%d N=%d K=%d)", M, N, K);
    TRID_REQUIRE(strideA % 4 == 0 && (b_mode == B_NC || strideB % 4 == 0), "trid_skinny_gemm_f32: batch strides of k-contiguous operands must be multiples of 4");
    TRID_REQUIRE(b_mode == B_KC || b_mode == B_NC, "trid_skinny_gemm_f32: b_mode is TRID_B_KC or TRID_B_NC");
    TRID_REQUIRE(K % 4 == 0 && lda % 4 == 0 && aligned16(A) && (b_mode == B_NC || (ldb % 4 == 0 && aligned16(B))),
                 "trid_skinny_gemm_f32: K, the k-contiguous row pitches must be multiples of 4 and the operands 16-byte aligned");
    TRID_REQUIRE((long long)M * lda * 4 < (1ll << 31) && (long long)(b_mode == B_KC ? N : K) * ldb * 4 < (1ll << 31),
                 "trid_skinny_gemm_f32: operands must stay below 2 GB (31-bit buffer offsets)");
    SkinnyParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias;
    p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.accumulate = accumulate;
    p.kslice = ((K + 7) / 8 + 63) / 64 * 64;  // (whole 64-deep loop trips; the tail reads zeros)
    p.sA = strideA; p.sB = strideB; p.sC = strideC; p.sBias = strideBias;
    const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 63) / 64), (unsigned)batch);
    if (b_mode == B_KC) hipLaunchKernelGGL(skinny_gemm_f32_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(skinny_gemm_f32_kernel<1>, grid, dim3(512), 0, (hipStream_t)stream, p);
    return check_launch("trid_skinny_gemm_f32");
}
