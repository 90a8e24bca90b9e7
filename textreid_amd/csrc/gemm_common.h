// Shared definitions of the GEMM / implicit-GEMM kernel family (gemm.hip, gemm_bf16.hip).
#pragma once
#include "common.h"

namespace trid {

typedef float v16f __attribute__((ext_vector_type(16)));

enum { A_KC = 0, A_MC = 1, A_CONV = 2 };
enum { B_KC = 0, B_NC = 1, B_CONV = 2 };

// Top-k admission filter epilogue (retrieval.hip; split kernel only): instead of storing C, every element
// >= thr[row * thr_stride] is appended to the row's candidate list (value, column) through an atomic
// per-row counter; `overflow` is raised when a list is full.
struct GemmFilter {
    const float* thr;
    int thr_stride;
    int* cnt;        // [M]
    float* cand;     // [M][cap][2]: value, column (int bits)
    int cap;
    int col0;        // added to the column index
    int* overflow;
};

// Eval-mode epilogues (BatchNorm of running statistics, ReLU, residual fused into the convolution: m_resnet.py:54-67 under
// model.eval()) write their output as a P16 tensor, so its fp16 scale must be fixed before the first element is known.
// It comes from a BOUND that needs no pass over anything:
//     |out[m][n]| <= |scale_n| * ||w_n||_1 * max|in| + |shift_n| (+ max|res|)  <=  coef[0] * tin + coef[1] (+ tres)
// with coef = (max_n |scale_n| ||w_n||_1, max_n |shift_n|) computed once per checkpoint (trid_eval_bound_coefs_f32) and
// tin / tres the TRUE maxima of the input / residual tensors, which their producers folded into device scalars
// (`tmax`, integer atomicMax) while writing them - the looseness of one layer's bound therefore never reaches the next.
// A bound 2^h above the true maximum costs nothing for elements within 2^-(16-h) of the maximum (two 11-bit planes,
// both normal) and leaves the others an absolute error of 2^-(38-h) of the maximum.
struct EvalBound {
    const float* coef;   // [2]
    const float* tin;    // true max|A| (device scalar)
    const float* tres;   // true max|res| or null
    float* out_bound;    // published by workgroup 0: the bound the output was scaled by (= the output tensor's amax scalar)
    float* out_tmax;     // atomicMax (bits of a non-negative float; zeroed by the caller): true max|out|
};

// BatchNorm-backward sums from the epilogue of the data-gradient GEMM that PRODUCES the gradient g (gemm_p16.hip, the
// staged-through-LDS store path): with y the saved conv output of the BatchNorm layer g belongs to, every 128-row tile adds
//   s1_c = sum g m,  s2_c = sum g m xhat,  max |g m|,  max |xhat|     (xhat = (y - mean_c) invstd_c, m = [scale_c y + shift_c > 0])
// over its rows and writes them in the layout bn_bwd_reduce_kernel leaves for bn_bwd_reduce_final_kernel (bn_pool.hip:
// [partial b][CW quads][8] sums, the same for the maxima; partial b = tile row * S + slice) - the reduce pass, which re-reads g
// and y from HBM only to form these sums, is then not run (m_resnet.py:54-67 backward).
struct BnBwdFuse {
    const float* y;        // [M][N] fp32, row pitch N; null: off
    const float* mean;     // [N]
    const float* invstd;
    const float* scale;
    const float* shift;
    float* ws;             // sums
    float* ws2;            // maxima
    int relu;              // 1: the layer's output went through ReLU (the mask is recomputed from y); 0: m = 1;
                           // 2: m = the bit of `mask` (a residual block's bn3: the ReLU came after the identity was added)
    const unsigned long long* mask;  // relu_mask words of bn_apply over [M][N] (the layout of GemmParams::cmask)
};

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long long lda, ldb, ldc;
    long long sA, sB, sC;  // batch strides (elements)
    int batch, splits;     // gridDim.z = batch * splits
    int k_chunk;           // K range per split (multiple of BK)
    long long sSplit;      // C offset per split (slab stride, elements)
    float alpha;
    int accumulate;
    const float* bias;  // [N] or null
    long long sBias;
    float* stats;       // [mblocks][N][2] (mean, M2) or null
    const float* res;   // [M][ldres] added after bias, or null
    long long ldres;
    int relu;           // clamp at zero last
    int H, W, Cin;      // conv geometry (A_CONV: M=Bimg*H*W,K=9*Cin; B_CONV: K=Bimg*H*W,N=9*Cin)
    FastDiv fdW, fdH, fdC;
    int mblocks, nblocks;
    GemmFilter filt;   // filt.thr == nullptr: normal epilogue
    const int* gate;   // optional device flag: the whole launch is a no-op when *gate == 0
    const float* a_amax;  // f16x3 arithmetic: device scalars holding max|A| / max|B| (null: operand used unscaled)
    const float* b_amax;
    int stats_w;          // floats per column in `stats`: 2 (mean, M2) or 4 (+ min, max: gemm_p16.hip)
    int c_fmt;            // gemm_p16.hip: 0 = C fp32, 2 = C plain bf16
    int wide_epilogue;    // gemm_p16.hip: stores (and accumulate / res reads) as whole rows through LDS
    int xcd_split;        // weight gradients (gemm_p16.hip): 1-D grid, every XCD owns whole K splits
    const unsigned long long* cmask;  // gemm_p16.hip, accumulate: bit per element of C (relu_mask layout) gating the OLD values
    // gemm_p16.hip, c_fmt == 1 (the eval-mode epilogue): C is written as a P16 tensor,
    //     out = act(colscale[n] * (A . B^T)[m][n] + bias[n] (+ res16[m][n])),
    // scaled by the bound eval_out_bound(ev) that every workgroup derives from device scalars before it stores anything
    const float* colscale;  // [N] or null
    const void* res16;      // P16 [M][N] (row pitch N) or null
    const float* res16_amax;
    EvalBound ev;
    int pool_w;             // > 0: the epilogue output is the 2x2 average pool of act(.) over images of this width ([M / 4][N])
    BnBwdFuse bb;
};

constexpr int BK = 32;

// split-precision variants (gemm_bf16.hip): fp32 operands split on the fly into 2 or 3
// bf16 planes, 3 or 6 bf16 MFMAs per product, fp32 accumulation
int gemm_bf16_dispatch(GemmParams& p, int a_mode, int b_mode, int precision, hipStream_t stream);

}  // namespace trid

namespace trid {
// retrieval (gemm_stream.hip): the admission-filter pass on pre-split operands, queries resident in registers
int stream_topk_filter(const void* g16, const float* g_amax, const void* q16, const float* q_amax, int G, int Q, const GemmFilter& filt,
                       hipStream_t stream);
}  // namespace trid

// trid_gemm_f32 plus the internal extras (either may be null).  With a filter the call returns
// TRID_E_UNSUPPORTED when the shape / precision does not run on the split kernel.
int trid_gemm_launch(const trid_gemm_desc* d, const trid::GemmFilter* filt, const int* gate, hipStream_t stream);

namespace trid {

}  // namespace trid
