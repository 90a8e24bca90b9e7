/* libtextreid_hip.so -- C ABI of the MI355X (gfx950) encode-and-match kernels.
 *
 * Drop-in boundary.  The reference (BrandonHanx/TextReID)
 * is pure Python on PyTorch and has no FFI of its own: every device kernel it
 * runs is dispatched implicitly by torch ops.  Each entry point below therefore
 * cites the reference call site (file:line under the reference root) whose
 * torch op(s) it replaces.  The Python host (textreid_amd/) binds these with
 * ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every buffer is DEVICE memory owned by the
 *    caller (PyTorch's caching allocator) and merely borrowed for the call;
 *  - asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*)
 *    and the call returns; no device synchronisation, no host reads of device data;
 *  - return 0 on success, negative TRID_E_* for argument errors, positive =
 *    hipError_t from the launch; trid_last_error_string() gives the message
 *    (thread-local);
 *  - re-entrant, no global mutable state;
 *  - activations are NHWC ("[B,H,W,C]" row-major), matrices row-major, fp32.
 */
#ifndef TEXTREID_HIP_H
#define TEXTREID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the entry points declared here are its whole dynamic symbol table */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define TRID_OK 0
#define TRID_E_INVALID (-1)
#define TRID_E_UNSUPPORTED (-2)

#define TRID_VERSION 100

int trid_version(void);
const char* trid_arch(void); /* "gfx950" */
const char* trid_last_error_string(void);

/* ------------------------------------------------------------------------- *
 * Dense contraction family (fp32 MFMA).  C[m,n] (+)= alpha*sum_k A(m,k)B(k,n) + bias[n]
 * Replaces: nn.Conv2d 1x1/3x3 fwd+bwd (m_resnet.py:18-27,41-47,161-170),
 * nn.Linear / F.linear (m_resnet.py:114-133, head.py:50-51,126-145), nn.GRU's
 * input/hidden projections (gru.py:36-43,76), torch.matmul / einsum
 * (losses.py:52-53,112; head.py:160-170; evaluation.py:120).
 * ------------------------------------------------------------------------- */
enum {
    TRID_A_KC = 0,   /* A[m*lda + k] */
    TRID_A_MC = 1,   /* A[k*lda + m] */
    TRID_A_CONV = 2, /* A = NHWC image [Bimg,H,W,Cin]; m = pixel, k = (tap,c); 3x3, stride 1, pad 1 */
    TRID_B_KC = 0,   /* B[n*ldb + k]  (weights [N,K]) */
    TRID_B_NC = 1,   /* B[k*ldb + n] */
    TRID_B_CONV = 2  /* B = NHWC image; k = pixel, n = (tap,c)  (weight gradient of a 3x3 conv) */
};

typedef struct trid_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int64_t strideA, strideB, strideC; /* batch strides, elements */
    int32_t batch;                     /* >= 1 */
    int32_t splits;                    /* split-K factor; split s writes C + s*strideSplit */
    int64_t strideSplit;
    int32_t a_mode, b_mode;
    float alpha;
    int32_t accumulate; /* C += ... */
    const float* bias;  /* [N] or NULL */
    int64_t strideBias; /* batch stride of bias, elements */
    float* stats;       /* NULL, or [ceil(M/128)][N][2] per-tile column (mean, M2) partials */
    int32_t H, W, Cin;  /* conv gather geometry */
    int32_t precision;  /* 0: exact fp32-input MFMA; 6 / 3: fp32 operands split on the fly into 3 / 2 bf16
                         * planes, 6 / 3 bf16 MFMAs per product, fp32 accumulate (6 is fp32-class, dropped
                         * terms <= 2^-26; 3 drops ~2^-17); 1: operands rounded to bf16, one MFMA per
                         * product, fp32 accumulate (bf16-autocast arithmetic); 16: fp32 operands scaled by a
                         * per-tensor power of two (a_amax / b_amax) and split into TWO fp16 planes (11 + 11
                         * significand bits, residual kept in the scaled domain), 3 fp16 MFMAs per product in one
                         * fp32 accumulator (representation + dropped term <= 3 * 2^-22 per product: fp32-class).
                         * Shapes the split kernel does not cover fall back to 0. */
    const float* residual; /* NULL, or [M][ldres] added after bias (eval: identity / folded downsample branch) */
    int64_t ldres;
    int32_t relu;          /* != 0: C = max(C, 0) last (eval: BatchNorm folded into weights + bias, ReLU here) */
    const float* a_amax;   /* precision 16 only: DEVICE scalars holding max|A| and max|B| over the whole operand */
    const float* b_amax;   /* (trid_amax_f32); NULL = operand used unscaled (must then lie in fp16's range)      */
    int32_t stats_minmax;  /* trid_gemm_p16 only, with stats: partials are [..][N][4] = (mean, M2, min, max) per column */
    int32_t c_format;      /* trid_gemm_p16 only: 0 = C is fp32; 2 = C is a plain bf16 tensor (ldc in elements; read as such
                            * with accumulate; BatchNorm partials are those of the ROUNDED values) - conv outputs and data
                            * gradients of the bf16 mode; needs batch == splits == 1 */
    const uint64_t* c_mask; /* trid_gemm_p16 only, with accumulate, batch == splits == 1, ldc == N, N % 4 == 0: NULL, or a bit per
                            * element of C in the layout of trid_bn_apply_*'s relu_mask; C = A . B^T + (bit ? C : 0).  The data
                            * gradient of a residual block's conv1 lands on dL/d(block output) masked by that output's ReLU
                            * (m_resnet.py:49-66 backward: out = relu(bn3(..) + identity)) without a masked copy of it */
    /* trid_gemm_p16 only, c_format == 1: the eval-mode epilogue (m_resnet.py:54-67 under model.eval(): BatchNorm of running
     * statistics, residual add and ReLU fused into the convolution).  C is written as a P16 tensor [M][N] (ldc == N, N % 32
     * == 0, batch == splits == 1, no accumulate / stats):
     *     C = act(col_scale[n] * alpha * (A . B^T)[m][n] + bias[n] (+ res_p16[m][n])),   act = relu ? max(., 0) : identity
     * scaled by the bound eval_coef[0] * *eval_tin + eval_coef[1] (+ *eval_tres) >= max|C|, which is published in *out_bound
     * (the output tensor's amax scalar); the TRUE max|C| is folded into *out_tmax (zeroed by the caller), the `eval_tin` of
     * the next layer.  eval_coef: trid_eval_bound_coefs_f32. */
    const float* col_scale;  /* [N] or NULL (1) */
    const void* res_p16;     /* P16 [M][N] or NULL */
    const float* res_amax;   /* the scalar res_p16 was packed with */
    const float* eval_coef;  /* device [2] */
    const float* eval_tin;   /* device scalar: true max|A| (any upper bound is valid) */
    const float* eval_tres;  /* device scalar: true max|res_p16|, NULL without a residual */
    float* out_bound;        /* device scalar, written */
    float* out_tmax;         /* device scalar, atomicMax'ed (bit pattern of a non-negative float) */
    int32_t eval_pool_w;     /* c_format 1, a_mode TRID_A_CONV, no residual: != 0 (= W) writes AvgPool2d(2) of act(.) instead: C = P16 [M / 4][N]
                              * (a stride-2 block's conv2 + bn2 + ReLU + avgpool, m_resnet.py:59-61); needs W | 128 with an even
                              * number of image rows per 128-row tile, H * W % 128 == 0, N > 64 */
    /* trid_gemm_p16, c_format 0, batch == splits == 1, ldc == N, N == 64 or N % 128 == 0 - a DATA GRADIENT that feeds a BatchNorm
     * backward (m_resnet.py:54-67): with bnb_y = the saved fp32 output y [M][N] of the convolution that BatchNorm layer
     * normalised (and its batch mean / invstd / scale / shift vectors [N]), every 128-row tile also writes the sums
     * sum g m, sum g m xhat and the maxima max|g m|, max|xhat| of its rows of C = g (m = [scale y + shift > 0] when bnb_relu == 1;
     * bnb_relu == 2: m = the bit of bnb_mask - the ReLU bit mask trid_bn_apply_p16_f32 wrote for the BLOCK OUTPUT that layer feeds
     * (a residual block's bn3: out = relu(bn3(.) + identity), m_resnet.py:62-66, where the sign of bn3's own output says
     * nothing) -; else 1) to bnb_ws / bnb_ws2 (trid_bn_bwd_fused_ws_floats floats each), in the layout trid_bn_bwd_final_f32 folds:
     * the BatchNorm-backward reduce pass over g and y is then not needed.  NULL: off. */
    const float* bnb_y;
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_scale;
    const float* bnb_shift;
    float* bnb_ws;
    float* bnb_ws2;
    int32_t bnb_relu;
    const void* bnb_mask;    /* bnb_relu == 2: ReLU bits over [M][N], the layout of c_mask */
} trid_gemm_desc;

int trid_gemm_f32(const trid_gemm_desc* d, void* stream);

/* C[M][ldc] (+)= alpha * A[M][lda] . B + bias[N] for M <= 128 rows (M = the batch): nn.Linear / F.linear of the attention
 * pool's q / c projections (m_resnet.py:114-133) and the embedding layers (head.py:50-51,126-129) and their data
 * gradients.  b_mode TRID_B_KC: B[N][ldb] (weights as stored); TRID_B_NC: B[K][ldb].  Exact fp32 MFMA; a workgroup owns
 * 32 columns, its 8 waves split the reduction, partial tiles folded in LDS in a fixed order (csrc/skinny_gemm.hip).
 * batch > 1: `batch` independent products at element strides strideA / strideB / strideC (/ strideBias) - the attention
 * pool's per-image and per-head products (32 heads x 193 tokens x 2048 channels per image). */
int trid_skinny_gemm_f32(const float* A, long long lda, const float* B, long long ldb, int b_mode, float* C, long long ldc,
                         const float* bias, int M, int N, int K, float alpha, int accumulate, int batch, long long strideA,
                         long long strideB, long long strideC, long long strideBias, void* stream);

/* ------------------------------------------------------------------------- *
 * P16: pre-split GEMM operands (no reference counterpart: PyTorch's fp32 convolutions need no operand format).
 * A [R rows][K] tensor (K % 32 == 0) in P16 holds x * 2^s (s from the tensor's amax, as precision 16 above) as TWO
 * fp16 planes, 4 bytes per element: row r, K group g -> 128 bytes at r*K*4 + g*128 = [hi(32 k) | lo(32 k)].
 * The kernel that PRODUCES an operand writes it split once; trid_gemm_p16 stages it with LDS-DMA (pure copies) and
 * evaluates hi*hi + hi*lo + lo*hi on the fp16 MFMA exactly as precision 16 does (same error bound).
 * ------------------------------------------------------------------------- */
/* fp32 [rows][K] (row pitch ldx elements) -> fmt 1: P16 (amax: device scalar max|x|, NULL: unscaled); fmt 2: plain
 * bf16 rows (round-to-nearest-even) */
int trid_p16_pack_f32(const float* x, long long rows, int K, long long ldx, const float* amax, void* out, int fmt, void* stream);
/* P16 -> fp32 [rows][K]: (hi + lo) / 2^s */
int trid_p16_unpack_f32(const void* in, long long rows, int K, const float* amax, float* out, void* stream);
/* w [N][T][C] fp32 -> P16 [C rows][K = T*N], k = t'*N + n, t' = flip ? T-1-t : t: the data-gradient operand of a
 * conv (autograd of nn.Conv2d, m_resnet.py:18-26): trid_weight_transpose_f32 + pack in one pass */
int trid_p16_pack_wt_f32(const float* w, int N, int T, int C, int flip, const float* amax, void* out, void* stream);
/* All conv filters of an encoder in one launch: table (device) = n_tensors x {src, dst, N, T, C, a} (int64), amax[a] =
 * max|tensor|; transposed == 0: trid_p16_pack_f32 of [N][T*C]; != 0: trid_p16_pack_wt_f32 (taps reversed for T > 1);
 * fmt 1: P16, fmt 2: plain bf16 rows (the operands of trid_gemm_p16 with precision 1; amax unused) */
int trid_p16_pack_multi_f32(const long long* table, const float* amax, int n_tensors, int transposed, int fmt, void* stream);
/* C = alpha * A . B^T (+ epilogues of trid_gemm_f32: bias, accumulate, residual, relu, split-K slabs, BatchNorm
 * partials) with A ([M][K], or an NHWC image for a_mode TRID_A_CONV) and B ([N][K]) in P16; lda / ldb = row pitch
 * in elements; a_amax / b_amax = the scalars the operands were packed with.  variant: tile shape; < 0 = the library's
 * choice (128 x 128 tiles of 8 waves; 9 = 96 x 128 tiles of 6 waves, an experiment kept for the record).  The BatchNorm
 * partials `stats` cover trid_gemm_p16_rows(M, N, precision, variant) rows each (128; 96 for variant 9). */
int trid_gemm_p16(const trid_gemm_desc* d, int variant, void* stream);
int trid_gemm_p16_rows(int M, int N, int precision, int variant);
/* 1: trid_gemm_p16 accepts the BatchNorm-backward sums (desc.bnb_y) for an [M, N] output on this build / in this environment
 * (whole 64- / 128-column tiles, the staged store path switched on: TRID_GEMM_WIDE_EPILOGUE) */
int trid_gemm_p16_bnb_ok(int M, int N);
/* The same product for SHORT reductions (K = 64 / 128 / 256: the expand 1x1 convolutions conv3 / downsample of layer1-3,
 * m_resnet.py:26,41-47, and the data gradients of conv1) as a streaming kernel (csrc/gemm_stream.hip): persistent
 * workgroups, the [32][K] filter panel of a wave in registers, activation tiles by LDS-DMA, stores straight from the
 * accumulators; bit-identical to trid_gemm_p16.  A P16 [M][K], B P16 [N][K] (N % 32 == 0), C fp32 [M][ldc] (NULL with stats: a statistics-only pass, nothing
 * is stored but the partials); accumulate: C += A . B^T; stats (may be NULL, not with accumulate): [ceil(M / rows)][N][4] = (mean, M2, min, max) per `rows` rows,
 * rows = trid_gemm_p16_stream_rows(M, N, K, accumulate) (0 there: shape not covered, use trid_gemm_p16).  c_mask (NULL, or
 * with accumulate, ldc == N and N % 256 == 0): as trid_gemm_desc.c_mask, C = A . B^T + (bit ? C : 0). */
int trid_gemm_p16_stream_rows(int M, int N, int K, int accumulate);
int trid_gemm_p16_stream_stats_rows(int M, int N, int K); /* rows per partial of the statistics-only pass (C == NULL) */
int trid_gemm_p16_stream(const void* A, const float* a_amax, const void* B, const float* b_amax, float* C, long long ldc,
                         float* stats, int M, int N, int K, int accumulate, const uint64_t* c_mask, void* stream);
/* conv3 + bn3 + identity + ReLU of an identity bottleneck block (m_resnet.py:57-66: out = relu(bn3(conv3(a)) + x)) in ONE
 * pass over the activations, once the BatchNorm statistics are known (a first trid_gemm_p16_stream pass with C = NULL and
 * stats given, then trid_bn_finalize_minmax_f32): the 1x1 convolution is recomputed - K = 64 / 128 input channels: 4 K bytes
 * per row - instead of storing y (4 N bytes) and reading it back.  A P16 [M][K], B P16 [N][K], res P16 [M][N] (scale from
 * res_amax), out P16 [M][N] with the scale of bound_a + bound_b (published in bound_sum), relu_mask (may be NULL) as
 * trid_bn_apply_p16_f32 writes it, y (may be NULL; fp32 [M][N]): the raw convolution output, kept when a backward pass
 * needs it.  Results are bit-identical to trid_gemm_p16_stream + trid_bn_apply_p16_f32.  _ok: shapes covered. */
int trid_conv1x1_bn_res_p16_ok(int M, int N, int K);
int trid_conv1x1_bn_res_p16(const void* A, const float* a_amax, const void* B, const float* b_amax, float* y,
                            const float* bn_scale, const float* bn_shift, const void* res, const float* res_amax,
                            void* out, const float* bound_a, const float* bound_b, float* bound_sum,
                            uint64_t* relu_mask, int M, int N, int K, int relu, void* stream);
/* eval mode (model.eval() of m_resnet.py:62-66): out = act(bn_scale * (A . B^T) + bn_shift (+ res)) as a P16 tensor in ONE pass,
 * the streaming kernel with the fused epilogue above; res (the identity / downsample branch, P16, may be NULL).  The output's
 * scale: the bound eval_coef[0] * *eval_tin + eval_coef[1] (+ *eval_tres), published in *out_bound; the true max|out| is
 * atomicMax'ed into *out_tmax (zeroed by the caller).  Shapes: trid_conv1x1_bn_res_p16_ok. */
int trid_conv1x1_eval_p16(const void* A, const float* a_amax, const void* B, const float* b_amax, const float* bn_scale,
                          const float* bn_shift, const void* res, const float* res_amax, void* out, const float* eval_coef,
                          const float* eval_tin, const float* eval_tres, float* out_bound, float* out_tmax, int M, int N, int K,
                          int relu, void* stream);
/* Weight gradients on P16 operands: C[M][N] = alpha * sum_k A[k][m] * B[k][n] with A = dL/dy [K pixels][M] and
 * B = the layer input [K pixels][N] (b_mode TRID_B_NC) or its 3x3 gather (TRID_B_CONV: N = 9*Cin, the NHWC image
 * [K pixels][Cin]); the K-major operands are transposed by the LDS read (ds_read_b64_tr_b16).  splits > 1 writes
 * split-K slabs (trid_slab_reduce_f32 folds them).  autograd of nn.Conv2d (m_resnet.py:18-26). */
int trid_gemm_p16_wgrad(const trid_gemm_desc* d, void* stream);

/* C[i] (+)= sum_s slab[s*strideSplit + i], i < n (n % 4 == 0) */
int trid_slab_reduce_f32(const float* slab, float* C, long long n, int splits, long long strideSplit,
                         int accumulate, void* stream);

/* w [N][T][C] -> wt [C][T][N]; flip != 0 reverses the tap order (3x3 dgrad = conv
 * with 180-degree rotated, transposed filters).  autograd of nn.Conv2d. */
int trid_weight_transpose_f32(const float* w, float* wt, int N, int T, int C, int flip, void* stream);

/* Stem conv1 (3x3, stride 2, pad 1, NCHW input, m_resnet.py:161-163,205) as
 * im2col: col[m][c*9+ky*3+kx], m = (b,yo,xo), row stride ldcol (>= Cin*9, padded
 * columns zero-filled). */
int trid_stem_im2col_f32(const float* img, float* col, int B, int Cin, int H, int W, int Ho, int Wo,
                         int ldcol, void* stream);

/* ---- the stem as bandwidth-shaped kernels (csrc/stem_conv.hip): every input pixel crosses the load path once.
 * Stem conv1 (nn.Conv2d(3, 32, 3, stride=2, padding=1, bias=False), m_resnet.py:161,205) straight from the NCHW image
 * batch [B][3][Hi][Wi] on the exact fp32 MFMA: y [B][Ho][Wo][32] (Ho = ceil(Hi/2), Wo = ceil(Wi/2)), w [32][27] as
 * stored; stats (may be NULL): [ceil(M/128)][32][4] = per-128-row (mean, M2, min, max) BatchNorm partials
 * (trid_bn_finalize_minmax_f32 with rows_per_part 128). */
int trid_stem_conv1_f32(const float* img, const float* w, float* y, float* stats, int B, int Hi, int Wi, void* stream);
/* eval mode (model.eval(): m_resnet.py:199-201 with BatchNorm on running statistics): the same kernel with the fused epilogue
 * out = act(bn_scale[c] * conv1(img) + bn_shift[c]) written as a P16 tensor [B][Ho][Wo][32]; scale / maximum scalars as in
 * trid_gemm_desc's eval fields (eval_tin = max|img|). */
int trid_stem_conv1_eval_p16(const float* img, const float* w, const float* bn_scale, const float* bn_shift, void* out,
                             const float* eval_coef, const float* eval_tin, float* out_bound, float* out_tmax, int B, int Hi, int Wi,
                             int relu, void* stream);
/* ... and its weight gradient (autograd of that nn.Conv2d): dw [32][27] = sum over output pixels of dy [B][Ho][Wo][32] (fp32)
 * times the image values under the 27 taps, gathered from the NCHW image (no im2col tensor), exact fp32 MFMA; slabs: scratch
 * of trid_stem_conv1_wgrad_slabs() * 864 floats (per-workgroup partial gradients, folded in a fixed order). */
/* Weight gradient of the same 3x3 convolutions (autograd of nn.Conv2d(C, N, 3, padding=1): stem conv2 / conv3, layer1's
 * conv2) with every pixel of x and dy staged once (rings of image rows in LDS, nine shifted transposing fragment reads): dy =
 * P16 NHWC [B][H][W][Cout], x = P16 NHWC [B][H][W][Cin], dw fp32 [Cout][9 * Cin] (column = tap * Cin + c, as
 * trid_gemm_p16_wgrad with TRID_B_CONV); slabs: scratch of trid_conv3x3_wgrad_halo_slabs() * Cout * 9 * Cin floats.
 * trid_conv3x3_wgrad_halo_rows: image rows per step, 0 = geometry not covered ((Cin, Cout) in {(32,32), (32,64), (64,64)},
 * W % 16 == 0) - use trid_gemm_p16_wgrad then. */
int trid_conv3x3_wgrad_halo_rows(int H, int W, int Cin, int Cout);
int trid_conv3x3_wgrad_halo_slabs(void);
int trid_conv3x3_wgrad_halo_p16(const void* dy, const float* dy_amax, const void* x, const float* x_amax, float* dw, float* slabs,
                                int B, int H, int W, int Cin, int Cout, void* stream);
int trid_stem_conv1_wgrad_slabs(void);
int trid_stem_conv1_wgrad_f32(const float* img, const float* dy, float* dw, float* slabs, int B, int Hi, int Wi, void* stream);
/* 3x3 / stride 1 / pad 1 convolution (nn.Conv2d(C, N, 3, padding=1, bias=False): stem conv2 / conv3,
 * m_resnet.py:165-170,206-207, and - with the transposed, 180-degree rotated filters of trid_p16_pack_multi_f32 - their
 * data gradients) for 32 / 64 channels on P16 operands: x = P16 NHWC [B][H][W][Cin] (scale from x_amax), w = P16
 * [Cout][9*Cin] (k = tap*Cin + c; w_amax), y fp32 [B][H][W][Cout].  A ring of image rows in LDS (LDS-DMA, zero padding
 * as data) feeds all nine taps; the filter slice of a wave lives in registers.  stats (may be NULL):
 * [B*H/rows][Cout][4] = (mean, M2, min, max) per step tile of rows*W pixels, rows = trid_conv3x3_halo_rows(H, W, Cin,
 * Cout) (= rows_per_part / W of trid_bn_finalize_minmax_f32); 0 there: the geometry is not covered, use trid_gemm_p16.
 * chunks_per_image: bands of rows an image is cut into (each band is walked by one persistent workgroup and re-reads
 * two halo rows); 0 = chosen so that the launch has >= 256 bands. */
int trid_conv3x3_halo_rows(int H, int W, int Cin, int Cout);
int trid_conv3x3_halo_p16(const void* x, const float* x_amax, const void* w, const float* w_amax, float* y, float* stats,
                          int B, int H, int W, int Cin, int Cout, int chunks_per_image, void* stream);
/* eval mode: the ring-of-rows kernel with the fused epilogue out = act(bn_scale[c] * conv(x, w) + bn_shift[c]) written as a
 * P16 tensor [B][H][W][Cout] (the stem's conv2, layer1's conv2 under model.eval(): m_resnet.py:22,57-60,202-204) - no fp32
 * output, no apply pass.  eval_coef / eval_tin / out_bound / out_tmax: as trid_gemm_desc's eval fields. */
int trid_conv3x3_halo_eval_p16(const void* x, const float* x_amax, const void* w, const float* w_amax, const float* bn_scale,
                               const float* bn_shift, void* out, const float* eval_coef, const float* eval_tin, float* out_bound,
                               float* out_tmax, int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream);
/* pool != 0 above: the output is the 2x2 average of act(.), P16 [B][H/2][W/2][Cout] - the stem's conv3 + bn3 + ReLU + AvgPool2d(2)
 * (m_resnet.py:205-207) in one kernel; covered geometries (32 -> 64 channels, W % 64 == 0): */
int trid_conv3x3_halo_eval_pool_ok(int H, int W, int Cin, int Cout);

/* ------------------------------------------------------------------------- *
 * BatchNorm2d (train: batch statistics + running update; eval: running stats),
 * ReLU, residual add, AvgPool2d(2).  m_resnet.py:19-29,38-49,57-66,164-171.
 * ------------------------------------------------------------------------- */
/* Merge per-tile (mean,M2) partials (rows_per_part rows each, last one ragged)
 * into batch mean / biased var; write mean, invstd, scale=gamma*invstd,
 * shift=beta-mean*scale; update running stats (unbiased var, momentum).
 * ws: NULL, or trid_bn_finalize_ws_bytes() bytes of device scratch private to the stream: with it the layers with
 * thousands of partials (>= 1024) run as two launches - range sums per channel octet over many workgroups, then a
 * merge in range order - instead of one workgroup per channel. */
long long trid_bn_finalize_ws_bytes(void);
int trid_bn_finalize_f32(const float* partials, int nparts, int rows_per_part, long long M, int C,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                         void* ws, void* stream);
/* eval mode: scale/shift from running statistics */
int trid_bn_eval_coeffs_f32(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* scale, float* shift, int C, void* stream);
/* eval mode (model.eval() of m_resnet.py:57-66, BatchNorm on running statistics): max|act(y*scale[c]+shift[c])| of a conv
 * output y from the [nparts][C][4] = (mean, M2, min, max) partials of the conv's epilogue, folded into *bound (device scalar,
 * zeroed by the caller, integer atomicMax): the fp16 scale of the P16 tensor the apply pass writes, known before it runs */
int trid_bn_eval_bound_f32(const float* partials, int nparts, int C, const float* scale, const float* shift, int relu,
                           float* bound, void* stream);
/* eval mode: per convolution i the pair coef[2i] = max_n |scale[n]| * sum_k |w[n][k]|, coef[2i+1] = max_n |shift[n]|, so that
 * |scale[n] * (w[n] . x) + shift[n]| <= coef[2i] * max|x| + coef[2i+1]: the output bound of the fused eval epilogues
 * (trid_gemm_desc.eval_coef, trid_conv1x1_eval_p16).  table (device) = n_tensors x {w, N, K, scale, shift} (int64; w fp32 with
 * each output channel's K weights contiguous); coef (device, 2 * n_tensors floats) must be zero on entry. */
int trid_eval_bound_coefs_f32(const long long* table, int n_tensors, float* coef, void* stream);
/* out = act( y*scale[c]+shift[c] + (res ? (rscale ? res*rscale[c]+rshift[c] : res) : 0) ).
 * relu_mask (optional, M*C/8 bytes): 1 bit per element, set where the pre-activation value is > 0, for
 * trid_bn_bwd_* mask_mode 3.  Element quad i = (row*C + c)/4 -> 64-bit words (i/64)*4 + (c%4), bit i%64. */
int trid_bn_apply_f32(const float* y, const float* scale, const float* shift, const float* res,
                      const float* rscale, const float* rshift, float* out, long long M, int C, int relu,
                      uint64_t* relu_mask, float* amax, void* stream);
/* out[b,y/2,x/2,c] = mean over 2x2 of act(y*scale+shift)   (scale==NULL: plain pooling of y).
 * amax (here, above and in trid_bn_bwd_apply_f32; may be NULL): device scalar, amax[0] = max(amax[0], max|out|) -
 * the producer of a GEMM operand hands the consumer its precision-16 scale without another pass. */
int trid_bn_apply_pool2_f32(const float* y, const float* scale, const float* shift, float* out, int B, int H,
                            int W, int C, int relu, float* amax, void* stream);
/* ---- the same passes PRODUCING / CONSUMING P16 tensors (pre-split GEMM operands, see trid_gemm_p16).  A P16 output
 * needs its scale before the pass runs: `bound*` are device scalars holding an upper bound of max|out| - exact for
 * act(BatchNorm(y)) from the column extremes of the conv epilogue (trid_bn_finalize_minmax_f32), a sum of two bounds
 * for a residual block output, a triangle-inequality bound for BatchNorm backward (trid_bn_bwd_reduce_bound_f32). */
/* trid_bn_finalize_f32 on (mean, M2, min, max) partials ([nparts][C][4], stats_minmax of trid_gemm_p16);
 * amax_out[0] = max(amax_out[0], max_c max|act(y_c*scale_c+shift_c)|) (relu != 0: act = ReLU), amax_out zeroed by the caller */
int trid_bn_finalize_minmax_f32(const float* partials, int nparts, int rows_per_part, long long M, int C,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                                int relu, float* amax_out, void* ws, void* stream);
/* Tensor formats of these entry points: 0 = fp32, 1 = P16, 2 = plain bf16 (configs[3]'s arithmetic: the convolutions
 * of the residual blocks read bf16 operands; written here with round-to-nearest-even, no scale, bounds unused). */
/* trid_bn_apply_f32 with the output in format fmt (1 / 2): P16 scaled for the bound bound_a[0] (+ bound_b[0] if not
 * NULL), bound_sum (may be NULL) receives that sum.  res_fmt: format of `res` (1: an identity residual, needs res_amax;
 * 2: a bf16 identity residual, or with rscale / rshift the bf16 raw output of the downsample convolution).  y_fmt: 0, or
 * 2 (bf16 mode: the conv output y is a bf16 tensor, written so by trid_gemm_p16 with c_format 2). */
int trid_bn_apply_p16_f32(const void* y, int y_fmt, const float* scale, const float* shift, const void* res, const float* rscale,
                          const float* rshift, int res_fmt, const float* res_amax, void* out, int fmt, long long M, int C,
                          int relu, uint64_t* relu_mask, const float* bound_a, const float* bound_b, float* bound_sum,
                          void* stream);
/* trid_bn_apply_pool2_f32 with the output in format fmt (P16: scale from bound[0]); in_fmt != 0: y itself is a tensor
 * of that format (plain pooling of a block input) */
int trid_bn_apply_pool2_p16_f32(const void* y, const float* scale, const float* shift, int in_fmt, const float* in_amax,
                                void* out, int fmt, int B, int H, int W, int C, int relu, const float* bound, void* stream);
/* dx[b,y,x,c] (+)= 0.25*g[b,y/2,x/2,c]; fmt: 0 = fp32 tensors, 2 = both plain bf16 (gradient tensors of the bf16 mode) */
int trid_avgpool2_bwd_f32(const void* g, void* dx, int B, int H, int W, int C, int accumulate, int fmt, void* stream);

/* BatchNorm backward.  g is dL/d(out).  mask_mode: 0 none, 1 recompute
 * (y*scale+shift > 0), 2 from `act` (> 0), 3 from the bit mask of trid_bn_apply_f32 passed as `act`.  pooled != 0: g has shape
 * [B,H/2,W/2,C] and the effective gradient is 0.25*g[b,y/2,x/2,c] (AvgPool2d(2)
 * after the activation).  Step 1 reduces dbeta = sum gm, dgamma = sum gm*xhat
 * (workspace: ws floats >= trid_bn_bwd_ws_floats(C)); step 2 writes
 * dy = scale*(gm - dbeta/M - xhat*dgamma/M) and optionally dres = gm. */
long long trid_bn_bwd_ws_floats(int C);
/* BatchNorm-backward sums written by the producing data-gradient GEMM (trid_gemm_desc.bnb_*): floats of EACH of its two
 * workspaces for M rows of C channels, and the fold that replaces trid_bn_bwd_reduce_bound_f32's: dgamma / dbeta [C] and
 * the bound of max|dy| (atomicMax'ed into *bound, zeroed by the caller) from the per-tile partials. */
long long trid_bn_bwd_fused_ws_floats(long long M, int C);
int trid_bn_bwd_final_f32(const float* ws, const float* ws2, long long M, int C, const float* scale, float* dgamma, float* dbeta,
                          float* bound, void* stream);
int trid_bn_bwd_reduce_f32(const float* g, const float* y, const float* act, const float* mean,
                           const float* invstd, const float* scale, const float* shift, int mask_mode,
                           int pooled, int B, int H, int W, int C, float* dgamma, float* dbeta, float* ws,
                           void* stream);
int trid_bn_bwd_apply_f32(const float* g, const float* y, const float* act, const float* mean,
                          const float* invstd, const float* scale, const float* shift, const float* dgamma,
                          const float* dbeta, int mask_mode, int pooled, int B, int H, int W, int C, float* dy,
                          float* dres, float* amax, void* stream);
/* trid_bn_bwd_reduce_f32 that also folds a bound of max|dy| into bound[0] (zeroed by the caller); ws as sized by
 * trid_bn_bwd_ws_floats */
int trid_bn_bwd_reduce_bound_f32(const float* g, const float* y, const float* act, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, int mask_mode, int pooled,
                                 int B, int H, int W, int C, float* dgamma, float* dbeta, float* ws, float* bound,
                                 void* stream);
/* trid_bn_bwd_reduce_f32 on an incoming gradient of format g_fmt: 0 = fp32, 2 = plain bf16 (configs[3]'s bf16 mode: the
 * gradient of a bf16 tensor is a bf16 tensor - written so by trid_gemm_p16 with c_format 2); y_fmt likewise for the conv
 * output y (in that mode a bf16 tensor too, as under autocast) */
int trid_bn_bwd_reduce_g_f32(const void* g, int g_fmt, const void* y, int y_fmt, const float* act, const float* mean,
                             const float* invstd, const float* scale, const float* shift, int mask_mode,
                             int pooled, int B, int H, int W, int C, float* dgamma, float* dbeta, float* ws,
                             void* stream);
/* trid_bn_bwd_apply_f32 with dy written in format fmt (1: P16 scaled for bound[0]; 2: bf16); g - and dres, its masked
 * copy - in format g_fmt (0 / 2) */
int trid_bn_bwd_apply_p16_f32(const void* g, int g_fmt, const void* y, int y_fmt, const float* act, const float* mean, const float* invstd,
                              const float* scale, const float* shift, const float* dgamma, const float* dbeta,
                              int mask_mode, int pooled, int B, int H, int W, int C, void* dy, int fmt, void* dres,
                              const float* bound, void* stream);

/* The two BatchNorm layers of a downsample block that share one incoming gradient (m_resnet.py:62-66: out = relu(bn3(.) +
 * downsample(x)): bn3 and the downsample branch's BatchNorm both see g masked by the block's ReLU bits) in ONE reduce pass
 * and ONE apply pass: g [M][C] fp32, relu_bits = the block output's relu_mask (trid_bn_apply_p16_f32), y1 / y2 the two saved
 * conv outputs with their batch (mean, invstd, scale) vectors.  dbeta (= sum g m) is the same for both layers and written to
 * dbeta and dbeta2; ws: 2 x trid_bn_bwd_ws_floats(C) floats; bound1 / bound2: zeroed device scalars that receive the bounds
 * of max|dy1| / max|dy2|, which the apply pass scales its P16 outputs by. */
int trid_bn_bwd_dual_reduce_bound_f32(const float* g, const uint64_t* relu_bits, const float* y1, const float* y2, const float* mean1,
                                      const float* invstd1, const float* scale1, const float* mean2, const float* invstd2,
                                      const float* scale2, long long M, int C, float* dgamma1, float* dgamma2, float* dbeta,
                                      float* dbeta2, float* ws, float* bound1, float* bound2, void* stream);
int trid_bn_bwd_dual_apply_p16_f32(const float* g, const uint64_t* relu_bits, const float* y1, const float* y2, const float* mean1,
                                   const float* invstd1, const float* scale1, const float* mean2, const float* invstd2,
                                   const float* scale2, const float* dgamma1, const float* dgamma2, const float* dbeta, long long M,
                                   int C, void* dy1, void* dy2, const float* bound1, const float* bound2, void* stream);

/* ------------------------------------------------------------------------- *
 * Attention pool (m_resnet.py:103-135), token-0 query only.
 * ------------------------------------------------------------------------- */
/* tok [B,ldt,C]: tok[b,0,:] = mean_t x[b,t,:] + pos[0]; tok[b,1+t,:] = x[b,t,:] + pos[1+t];
 * rows T+1..ldt-1 are zero padding (ldt multiple of 4 keeps the token axis float4-addressable) */
int trid_attnpool_tokens_f32(const float* x, const float* pos, float* tok, int B, int T, int C, int ldt,
                             void* stream);
/* the same with x in the format the last residual block wrote it (x_fmt 0: fp32, 1: P16 with its amax scalar, 2: plain
 * bf16) - no unpack pass - and the token loop spread over the workgroup */
int trid_attnpool_tokens_fmt_f32(const void* x, int x_fmt, const float* x_amax, const float* pos, float* tok, int B, int T,
                                 int C, int ldt, void* stream);
/* dx[b,t,:] = dtok[b,1+t,:] + dtok[b,0,:]/T ; dpos[t,:] = sum_b dtok[b,t,:] */
int trid_attnpool_tokens_bwd_f32(const float* dtok, float* dx, float* dpos, int B, int T, int C, int ldt,
                                 void* stream);
/* row softmax over the first n columns of each row (ld >= n), in place; pad columns zeroed */
int trid_softmax_rows_f32(float* s, long long rows, int n, int ld, void* stream);
/* ds = p * (dp - sum_j p_j dp_j), row-wise; writes into ds (may alias dp) */
int trid_softmax_rows_bwd_f32(const float* p, const float* dp, float* ds, long long rows, int n, int ld,
                              void* stream);
/* out[n] (+)= sum_m x[m*ld + n] */
int trid_colsum_f32(const float* x, float* out, long long M, int N, long long ld, int accumulate, void* stream);

/* ------------------------------------------------------------------------- *
 * Text encoder (gru.py:48-82): table gather, masked BiGRU cell, max over time.
 * ------------------------------------------------------------------------- */
/* x[b*L + t, :] = table[tokens[b*ldtok + t], :]  for t < L */
int trid_embedding_gather_f32(const float* table, const int64_t* tokens, float* x, int B, int L, int ldtok, int E,
                              long long vocab, void* stream);
/* Gradient of a trainable token-embedding table (gru.py:23-24, `use_onehot == "yes"`: nn.Embedding(vocab, embed, padding_idx=0)):
 * dtable[v] = sum over the positions whose token is v of dX[position] ([B*L][E], position = b * L + t), row padding_idx zero.
 * Deterministic (no sort, no floating-point atomics): the first position of a token owns its row and adds in position order. */
int trid_embedding_bwd_f32(const float* dX, const int64_t* tokens, int B, int L, int ldtok, int E, float* dtable,
                           long long vocab, long long padding_idx, void* stream);
/* One time step of both directions.  s = step index; direction d processes
 * t = s (d=0) or t = Lmax-1-s (d=1).  gi: [B*L, 2*3H] input projections (row
 * b*L+t, col d*3H + gate*H + j); gh: [2,B,3H]; h: [2,B,H] updated in place;
 * saved gates (r,z,n,hn) -> gates [B,4H] per direction at gates + d*gates_dstride,
 * previous state -> hprev [B,H] at hprev + d*hprev_dstride (NULL to skip, e.g.
 * the key encoder); running max over time in
 * maxv/argt [B,2H] (column d*H+j). */
int trid_gru_cell_fwd_f32(const float* gi, const float* gh, float* h, const int64_t* lengths, float* gates,
                          float* hprev, float* maxv, int32_t* argt, int s, int Lmax, int L, int B, int Hd,
                          long long gates_dstride, long long hprev_dstride, void* stream);
/* maxv/argt init: 0/-1 when length < batch maximum (a zero pad row enters the max, gru.py:63) else -inf/-1; the batch
 * maximum is Lmax, or lmax_dev[0] (device scalar) when that pointer is not NULL and Lmax is only an upper bound */
int trid_gru_max_init_f32(float* maxv, int32_t* argt, const int64_t* lengths, int Lmax, const int64_t* lmax_dev, int B,
                          int Hd, void* stream);
/* Backward of one step (reverse order of s).  dh [2,B,H] carries dL/dh; adds the
 * max-pool gradient dout[b, d*H+j] where argt == t; writes dgi rows into dGi
 * [B*L, 2*3H], dgh [2,B,3H]; dh <- dh*z (+ pass-through when inactive). */
int trid_gru_cell_bwd_f32(const float* dout, const int32_t* argt, const float* gates, const float* hprev,
                          const int64_t* lengths, float* dh, float* dGi, float* dgh, int s, int Lmax, int L,
                          int B, int Hd, long long gates_dstride, long long hprev_dstride, long long dgh_dstride,
                          void* stream);

/* --- One launch per time step (gru_step.hip): recurrent product + gates + state + running max fused, both
 * directions, fp32-class arithmetic (fp16 two-plane split, 3 MFMA products).  Replaces the trid_gemm_f32 +
 * trid_gru_cell_* pair of a step (gru.py:66-82); built for H % 32 == 0, 32 <= H <= 768, else TRID_E_UNSUPPORTED.
 * trid_gru_pack_whh_f16: w_hh [2, 3H, H] (forward, reverse) -> the forward and backward fragment images (each
 * trid_gru_whh_image_bytes(H) bytes, 0 = unsupported H), once per pass; w_amax = device scalar max|w_hh|
 * (trid_amax_f32), also handed to every step.
 * Forward step s: hp_in / hp_out [2, Bp, H] uint32 = the hidden state entering / leaving the step as packed fp16
 * planes ((hi | lo << 16) of h * 2^13; Bp = B rounded up to 16, padding rows zero; all-zero before step 0; the
 * caller alternates two buffers); everything else as trid_gru_cell_fwd_f32 (h [2,B,H] in place, gates / hprev
 * slices of this step or NULL, maxv / argt).
 * Backward step s (descending): dgh_in = the [2,B,3H] slice written by step s+1 and amax_in = the per-workgroup
 * maxima it published (both NULL at the first processed step); amax_out receives this step's: both are
 * trid_gru_step_workgroups(B, H) floats;
 * dh [2,B,H] carries dL/dh in place: dh <- (dh + dgh_in @ W_hh [+ dout where argt == t]) * z; the rest as
 * trid_gru_cell_bwd_f32. */
long long trid_gru_whh_image_bytes(int H);
int trid_gru_step_workgroups(int B, int H);
int trid_gru_pack_whh_f16(const float* w_hh, const float* w_amax, void* img_fwd, void* img_bwd, int H, void* stream);
int trid_gru_step_fwd_f32(const void* img_fwd, const float* w_amax, const void* hp_in, void* hp_out, float* h,
                          const float* gi, const int64_t* lengths, float* gates, float* hprev, float* maxv,
                          int32_t* argt, int s, int Lmax, int L, int B, int Bp, int H, long long gates_dstride,
                          long long hprev_dstride, void* stream);
int trid_gru_step_bwd_f32(const void* img_bwd, const float* w_amax, const float* dgh_in, const float* amax_in,
                          float* amax_out, const float* dout, const int32_t* argt, const float* gates,
                          const float* hprev, const int64_t* lengths, float* dh, float* dGi, float* dgh_out, int s,
                          int Lmax, int L, int B, int H, long long gates_dstride, long long hprev_dstride,
                          long long dgh_dstride, void* stream);

/* ------------------------------------------------------------------------- *
 * Embedding head and losses (head.py:126-175, losses.py, moco_head/loss.py).
 * ------------------------------------------------------------------------- */
/* y = x / max(||x||_2, eps) row-wise; inv_norm[rows] saved */
int trid_l2norm_rows_f32(const float* x, float* y, float* inv_norm, long long rows, int C, float eps, void* stream);
/* dx = (dy - y * <dy,y>) * inv_norm  (+ dx if accumulate) */
int trid_l2norm_rows_bwd_f32(const float* dy, const float* y, const float* inv_norm, float* dx, long long rows,
                             int C, int accumulate, void* stream);
/* flag[k] = 1 if id_queue[k] equals any ids[i] (head.py:148-157: one shared column set per batch) */
int trid_queue_hit_mask(const int64_t* id_queue, const int64_t* ids, uint8_t* flag, int K, int B, void* stream);
/* InfoNCE over [pos | masked negs]/T with label 0 (losses.py:206-217).
 * S: [B, ldS] similarity of queries vs the K queue rows (raw dot products), in
 * place becomes dL/dS (already scaled by gscale/(B*T)); pos[b] = <q_b, key_b>;
 * outputs loss_rows[b] = lse - pos/T and dpos[b].  Rows are cut into 4096-column segments (one workgroup
 * each, two passes) so that long queues fill the chip; ws floats >= trid_infonce_ws_floats(B, K). */
long long trid_infonce_ws_floats(int B, int K);
int trid_infonce_rows_f32(float* S, const float* pos, const uint8_t* hit, float* loss_rows, float* dpos, int B,
                          int K, int ldS, float invT, float gscale, float* ws, void* stream);
/* Same, for the fused queue path (head.py:159-170): the positive logit <q_b, key_b> is formed inside (q, key
 * [B, C]) and dq0[b,:] = dL/dpos[b] * key[b,:] is written, onto which the caller accumulates dL/dS @ queue. */
int trid_infonce_queue_rows_f32(float* S, const float* q, const float* key, const uint8_t* hit, float* loss_rows,
                                float* dq0, int B, int K, int ldS, int C, float invT, float gscale, float* ws,
                                void* stream);
/* Fused queue similarity + masked InfoNCE of BOTH modalities, forward and gradient, in ONE pass over the two
 * [K, C] row-major queues (head.py:148-170 + losses.py:206-217; no [B, K] matrix exists anywhere):
 *   modality 0: image queries v_q against t_queue, positive key t_key (head.py:160-164);
 *   modality 1: text  queries t_q against v_queue, positive key v_key (head.py:166-170).
 * Queries and queue rows must be L2-normalised (head.py:128-129,140,145), |<q, k>| <= logit_bound (1 for unit
 * vectors): the kernel uses the fixed shift logit_bound/T instead of a running maximum.  A queue row k is
 * filtered when id_queue[k] equals ANY ids[i], i < B (one shared column set per batch, head.py:148-157).  Outputs loss_rows[2][B] = lse - pos/T and dq[2][B][C] = dL/dq for
 * L = gscale * mean_b(loss_rows) summed over the modalities.  precision: 6 = fp32-class (fp16 two-plane split with fixed
 * scales, 6 MFMA products per query x row x channel), 3 = fp32-class on three bf16 planes (11 products), 1 = bf16 operands.  nwg_hint: workgroups per modality (0 = default).  Built for C = 256 and
 * K % 32 == 0, otherwise TRID_E_UNSUPPORTED (the caller then takes trid_gemm_f32 + trid_infonce_queue_rows_f32).
 * ws floats >= trid_queue_nce_ws_floats(B, K, C, nwg_hint) (0 = unsupported shape).  Bit-reproducible: partial
 * sums are folded in a fixed order, no floating-point atomics.
 * ticket / loss (both or NULL): with a device word that holds ZERO on entry (one fresh word per call; it is left at
 * 2 B) the finish launch also writes loss[0] = loss_scale * sum(loss_rows) - the mean over the batch of losses.py:216-217
 * with loss_scale = 1 / B - folded in index order by whichever workgroup finishes last: two launches for the block. */
long long trid_queue_nce_ws_floats(int B, int K, int C, int nwg_hint);
int trid_queue_nce_f32(const float* v_q, const float* t_q, const float* v_key, const float* t_key,
                       const float* t_queue, const float* v_queue, const int64_t* id_queue, const int64_t* ids,
                       float* loss_rows, float* dq, int B, int K, int C, float invT, float logit_bound, float gscale,
                       int precision, int nwg_hint, float* ws, unsigned int* ticket, float* loss, float loss_scale,
                       void* stream);
/* Largest magnitudes as device scalars (operand scales of trid_gemm_desc.precision == 16; no reference
 * counterpart: PyTorch's fp32 convolutions need no range management).
 * trid_amax_f32: out[0] = max(out[0], max|x|) - `out` must hold 0 (or an earlier partial maximum) on entry.
 * trid_amax_multi_f32: out[t] = max(out[t], max|tensor t|) for a device table of n_tensors pointers / element
 * counts (out zeroed by the caller). */
int trid_amax_f32(const float* x, long long n, float* out, void* stream);
int trid_amax_multi_f32(const float* const* ptrs, const long long* sizes, int n_tensors, float* out, void* stream);
/* Device-side image input pipeline (lib/data/transforms.py:4-43): B raw uint8 HWC images of arbitrary
 * sizes (concatenated in `src`, byte offsets `offset[B]`, sizes `hw[B][2]`) -> out fp32 [B,3,H,W]:
 * Resize((H,W)) exactly as Pillow's antialiased BILINEAR (two fixed-point passes rounding to uint8; the per-image
 * weight tables xbounds[B][W][2] / xweights[B][W][KX] / ybounds[B][H][2] / yweights[B][H][KY] are
 * Resample.c's precompute_coeffs + normalize_coeffs_8bpc), horizontal flip, zero Pad(pad) + crop at (top, left),
 * /255, (x - mean) / std, erase rectangle filled with `erase` values.  params[B][8] = {flip, crop_top, crop_left,
 * erase_i, erase_j, erase_h, erase_w, 0} (device); mean3_std3_erase3_host = 9 HOST floats.  ws bytes >=
 * trid_image_pipeline_ws_bytes(B, max source height, W) holds the horizontally resampled rows. */
long long trid_image_pipeline_ws_bytes(int B, int max_src_h, int W);
int trid_image_pipeline_u8(const uint8_t* src, const long long* offset, const int* hw, const int* xbounds,
                           const int* xweights, const int* ybounds, const int* yweights, const int* params, int B,
                           int H, int W, int KX, int KY, int max_src_h, int pad, const float* mean3_std3_erase3_host,
                           uint8_t* ws, float* out, void* stream);
/* dx = act > 0 ? dy : 0 - backward of the nn.ReLU inside the MOCO.FC projection heads (head.py:33-42) */
int trid_relu_bwd_f32(const float* dy, const float* act, float* dx, long long n, void* stream);
/* rowdot[b] = <x_b, y_b> */
int trid_rowdot_f32(const float* x, const float* y, float* out, long long rows, int C, void* stream);
/* dx[b,:] (+)= s[b]*y[b,:] */
int trid_rowscale_add_f32(const float* s, const float* y, float* dx, long long rows, int C, int accumulate,
                          void* stream);
/* Label-smoothed CE rows (losses.py:6-39 via :42-62).  logits [rows, ld] (first
 * n valid); in place -> dlogits scaled by gscale/rows_per_loss; loss_rows[r] =
 * -(1-eps)*logp[label] - eps/n * sum_j logp_j. */
int trid_smooth_ce_rows_f32(float* logits, const int64_t* labels, float* loss_rows, long long rows, int n, int ld,
                            float epsilon, float gscale, void* stream);
/* projection [C, N] -> column-normalised copies pn [C, ldn] (optional, may be NULL) and
 * pnt [ldn, C] (class-major, zero rows for the ldn-N padding); inv_norm[N] */
int trid_colnorm_f32(const float* proj, float* pn, float* pnt, float* inv_norm, int C, int N, int ldn, void* stream);
/* dproj[c,j] = (dpnt[j,c] - pnt[j,c]*<dpnt[j,:],pnt[j,:]>) * inv_norm[j] */
int trid_colnorm_bwd_f32(const float* dpnt, const float* pnt, const float* inv_norm, float* dproj, int C, int N,
                         int ldn, void* stream);
/* Global-align loss elementwise part (losses.py:102-128): S [B, ldS] cosine
 * matrix in place -> dL/dS (scaled by gscale); loss_rows[b] = row sum of the
 * softplus terms * 2/B. */
int trid_global_align_rows_f32(float* S, const int64_t* ids, float* loss_rows, int B, int ldS, float alpha,
                               float beta, float scale_pos, float scale_neg, float gscale, void* stream);
/* out = g3[0]*a + g3[1]*b + g3[2]*c, g3 device-resident upstream gradients (b, c may be NULL) */
int trid_axpby3_f32(float* out, const float* a, const float* b, const float* c, const float* g3, long long n,
                    void* stream);
/* out[0] (+)= scale * sum_i x[i] */
int trid_sum_f32(const float* x, float* out, long long n, float scale, int accumulate, void* stream);

/* ------------------------------------------------------------------------- *
 * MoCo state: momentum update, enqueue (head.py:73-109), optimiser.
 * ------------------------------------------------------------------------- */
/* Multi-tensor k = k*m + q*one_minus_m (two rounded products + one rounded sum, the
 * reference's op order; one_minus_m is passed so the host can form 1-m in double as
 * Python does).  ptr tables are DEVICE arrays of n_tensors addresses; chunk table:
 * chunk c covers elements [chunk_off[c], +chunk_len) of tensor chunk_tensor[c]. */
int trid_ema_multi_f32(const uint64_t* k_ptrs, const uint64_t* q_ptrs, const int64_t* sizes,
                       const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_len, float m,
                       float one_minus_m, void* stream);
/* Multi-tensor Adam / AdamW step (torch.optim.Adam semantics, lib/solver/build.py:6-40:
 * per-tensor lr and weight decay).  bias_c1[i] = 1-b1^t_i, bias_c2[i] = sqrt(1-b2^t_i): DEVICE arrays, one
 * entry per tensor (torch.optim.Adam keeps one step count t_i per parameter); step_size = lr/bias_c1. */
int trid_adam_multi_f32(const uint64_t* p_ptrs, const uint64_t* g_ptrs, const uint64_t* m_ptrs,
                        const uint64_t* v_ptrs, const int64_t* sizes, const float* lrs, const float* wds,
                        const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_len,
                        float beta1, float beta2, float eps, const float* bias_c1, const float* bias_c2, int decoupled,
                        void* stream);
/* Ring-buffer push at device-resident pointer: queue row-major [K, C]
 * (transpose of the reference's [C,K], so the push is one contiguous slab). */
int trid_enqueue_f32(float* v_queue, float* t_queue, int64_t* id_queue, int64_t* ptr, const float* v_keys,
                     const float* t_keys, const int64_t* ids, int K, int C, int B, void* stream);

/* ------------------------------------------------------------------------- *
 * Retrieval (evaluation.py:11-37,117-120): per-query top-k over a gallery shard.
 * ------------------------------------------------------------------------- */
/* sim = q @ g.T for q [Q,C], g [G,C] (both L2-normalised by the caller), fused
 * per-row top-k (k <= 16) sorted descending, ties -> lower index first.
 * out_val [Q,k] f32, out_idx [Q,k] i64 (+ idx_offset).  ws floats >= trid_topk_ws_floats(Q,G,k).
 * The first 8192 gallery rows go through a [Q, 8192] similarity panel and a streaming row scan; the rest
 * through ONE GEMM whose epilogue keeps only elements that reach the row's current k-th value (exact for
 * any input order: list overflow falls back to panel passes on device, without a host round trip).
 * precision: arithmetic of the similarity GEMM, as trid_gemm_desc.precision (16: fp16 two-plane split, fp32-class,
 * needs q_amax / g_amax = device scalars max|q|, max|g| (trid_amax_f32; NULL -> runs as 6); 6: split bf16,
 * fp32-class; 0: exact fp32-input MFMA). */
long long trid_topk_ws_floats(int Q, int G, int k);
int trid_sim_topk_f32(const float* q, const float* g, float* out_val, int64_t* out_idx, int Q, int G, int C,
                      int k, long long idx_offset, int precision, const float* q_amax, const float* g_amax, float* ws, void* stream);
/* The same with the operands ALSO given pre-split (C == 256): q16 = P16 [ceil32(Q)][256] (padding rows zero), g16 = P16 [G][256],
 * packed with q_amax / g_amax (trid_p16_pack_f32).  The gallery embeddings are written once and scored against every query
 * panel: split once instead of once per panel, and the admission-filter pass runs on the streaming kernel with 256 queries
 * resident in a workgroup's registers (csrc/gemm_stream.hip) - the gallery crosses LDS once per 256 queries, nothing is
 * stored but candidates.  Results as trid_sim_topk_f32 with precision 16 (same arithmetic; values agree to the last bits).
 * The rows beyond the first panel are filtered in SEGMENTS of growing length (x 8: 8192 | 65536 | 524288 | ...), each merged into the
 * running top-k before the next starts, so a query's threshold is the k-th best of everything before the segment (~7 k
 * candidates per query and segment); ws also keeps the top-k as it stood after the first panel, which the fall-back restores.
 * mode 0: as trid_sim_topk_f32 (the overflow fall-back passes are enqueued behind the fused pass, gated on a device flag).
 * mode 1: without them - the caller reads the int at ws[trid_topk_ws_flag_offset(Q, G)] afterwards and, when it is non-zero (a
 * candidate list overflowed: an adversarially ordered gallery), calls mode 2 with the same arguments and ws, which redoes the
 * columns beyond the first panel densely.  One 4-byte read instead of ~2 G / 8192 gated no-op launches (1.4 of 17 ms at G = 1e6). */
int trid_sim_topk_p16(const float* q, const float* g, const void* q16, const void* g16, float* out_val, int64_t* out_idx, int Q, int G,
                      int k, long long idx_offset, const float* q_amax, const float* g_amax, float* ws, int mode, void* stream);
long long trid_topk_ws_flag_offset(int Q, int G);

/* per-row top-k of a given similarity matrix (rank(get_mAP=False), evaluation.py:17-19) */
int trid_topk_rows_f32(const float* sim, int ld, int Q, int G, int k, float* out_val, int64_t* out_idx,
                       void* stream);
/* per-row full descending argsort (rank(get_mAP=True), evaluation.py:14), ties -> lower column first.  G <= 16384:
 * one in-LDS bitonic sort per row, no workspace (ws may be NULL).  Larger rows: packed keys + rocPRIM segmented radix
 * sort in the caller's workspace of trid_argsort_ws_bytes(Q, G) bytes (256-byte aligned; 0 = none needed, or Q*G >= 2^31:
 * sort the rows in batches). */
long long trid_argsort_ws_bytes(int Q, int G);
int trid_argsort_rows_desc_f32(const float* sim, int ld, int Q, int G, int64_t* out_idx, void* ws, long long ws_bytes,
                               void* stream);
/* matches = g_pids[indices] == q_pids; first_hit[q] = rank of the first match (INT_MAX if none);
 * ap[q] = average precision over the R ranked items (NaN without relevant items);
 * cmc[t] = 100*mean(first_hit < topk[t])  (evaluation.py:20-36) */
int trid_rank_metrics(const int64_t* indices, const int64_t* q_pids, const int64_t* g_pids, int Q, int R,
                      int32_t* first_hit, float* ap, const int64_t* topk, int ntopk, float* cmc, void* stream);

/* k-reciprocal re-rank term (evaluation.py:40-65): out[i,j] = alpha*jaccard(qnn[i,:k], gnn[j,:k]) + base[i,j]
 * (base may be NULL); qnn [Q,k], gnn [G,k] top-k neighbour indices, k <= 8 */
int trid_jaccard_add_f32(const int64_t* qnn, const int64_t* gnn, const float* base, long long ldb, float* out, int Q,
                         int G, int k, float alpha, void* stream);

/* ------------------------------------------------------------------------- *
 * The train step as one host call (lib/engine/trainer.py:72-91: model(images, captions) -> sum of the losses ->
 * zero_grad -> backward -> optimizer.step, ~1100 launches on four streams).
 * ------------------------------------------------------------------------- */
/* graph: the hipGraph_t a stream capture of one step produced (kept alive by the caller for as long as the handle lives: the
 * launches point at the graph's own argument copies).  build reads its kernel / memcpy / memset nodes and edges back and lays
 * them out on at most max_lanes streams: a node continues the stream of a predecessor it directly follows, every other edge
 * becomes an event.  run re-issues the step as ordinary stream launches - what the eager step does, without its 35-39 ms of
 * Python / ctypes time per step and with every stream fed at once; work enqueued on `origin_stream` before the call happens
 * before the step, work enqueued after it happens after.  Graphs with other node types (host callbacks, child graphs) are
 * refused with TRID_E_UNSUPPORTED.  info: counts[8] = nodes, kernels, copies, memsets, lanes, events, waits, empty nodes. */
int trid_step_replay_build(void* graph, int max_lanes, void** out_handle);
int trid_step_replay_info(void* handle, int* counts);
int trid_step_replay_run(void* handle, void* origin_stream);
int trid_step_replay_destroy(void* handle);
/* Data parallel (train_net.py:50-56 around trainer.py:72-91): the step's collectives are not recorded.  trid_step_marker enqueues the
 * empty marker kernel that stands for collective `id` in the recording (its stream and edges are the collective's); a plan built
 * from a recording with markers is replayed in segments: run_segment issues the nodes up to the next marker and returns 1 with the
 * collective's id and the stream the host must enqueue it on (through the communication library's own launch path), then is called
 * again; it returns 0 once the step is complete.  A launch failure in mid-step joins what was issued back into origin_stream and
 * poisons the handle (every later call fails).  markers: number of cut points of the plan. */
int trid_step_marker(int id, void* stream);
int trid_step_replay_markers(void* handle);
int trid_step_replay_run_segment(void* handle, void* origin_stream, int* marker_id, void** lane_stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* TEXTREID_HIP_H */
