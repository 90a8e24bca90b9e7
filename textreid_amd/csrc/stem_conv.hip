// The stem's convolutions (m_resnet.py:161-170, 199-207) as bandwidth-shaped kernels.
//
// The stem works on the largest images of the network (B x 192 x 64 pixels at 384 x 128 input) with the fewest channels
// (3 -> 32 -> 32 -> 64): every convolution is HBM-bound by construction (conv2: 201 MB in, 201 MB out, 29 GFLOP), so a
// GEMM-shaped kernel that stages one K tile per tap re-reads every activation nine times through the texture path and
// never gets near the memory rate.  Here each input pixel crosses the load path ONCE:
//
//   * conv3x3_halo_p16_kernel (conv2, conv3, and their data gradients): the input is a P16 NHWC tensor (gemm_p16.hip: a
//     pixel's 32 channels are one 128-byte line = [hi x 32 | lo x 32] fp16).  A persistent workgroup walks down a band of
//     image rows with a RING of image rows in LDS: each step brings TH new rows in by LDS-DMA (zero columns left and
//     right, zero rows above / below the image: the padding of the convolution is data in LDS, not masks in the loop),
//     computes TH output rows from the TH + 2 rows it has, and stores them.  The nine taps are nine shifted ds_read_b128
//     fragment reads of the SAME LDS image (16-byte units XOR-swizzled by the pixel index: conflict-free for any shift).
//     The filters never touch LDS: a wave owns one (32 output channels x 32 input channels) slice, whose 9 x 2 x 2
//     MFMA B fragments live in 144 VGPRs for the whole launch.  Same arithmetic as gemm_p16_kernel<A_CONV> (products
//     hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_f16, same order: bit-identical for 32 input channels).
//     Epilogue: BatchNorm partials (mean, M2, min, max) per step tile, Chan-merged over the waves of a tile.
//   * stem_conv1_kernel: the 3x3 / stride 2 convolution of the 3-channel image, straight from the NCHW input (no im2col
//     tensor: 176 MB written and read back per pass before) on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32, K = 27).

#include <algorithm>
#include <mutex>

#include "split_common.h"

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
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& rs, void* lds_base, unsigned voffset, bool nt) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    if (nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_base, 16, voffset, 0, 0, 2);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_base, 16, voffset, 0, 0, 0);
}

// LDS stores the compiler does not see: beside an LDS-DMA in flight hipcc orders every ds_write behind `s_waitcnt vmcnt(0)`
// (the DMA is a pending LDS write on the VM counter and it cannot prove the two do not alias), which would serialise the
// epilogue behind the prefetch of the next rows.  The scratch regions written here are disjoint from the ring.
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(uintptr_t)p; }
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_store4(const void* p, float4 v) {
    const v4f r = {v.x, v.y, v.z, v.w};  // (a 128-bit register tuple: the HIP float4 struct is not an asm operand)
    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_off(p)), "v"(r) : "memory");
}
__device__ __forceinline__ void lds_store1(const void* p, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_off(p)), "v"(v) : "memory");
}

// Eval-mode epilogue of the kernels below, whose accumulator layout gives a lane ONE output channel (lane & 31 of a 32-channel
// group = one 128-byte P16 line per pixel: [hi x 32 | lo x 32] fp16) and 16 pixels: v = act(acc * scale + shift) is split into
// its two fp16 planes in the lane, neighbouring lanes exchange one plane each (channel pairs make a dword), and every lane
// stores ONE dword per pixel - even lanes the high parts of channels (n, n + 1), odd lanes the low parts of (n - 1, n): the
// same 16 stores per lane as the fp32 form, 32 lanes = one full line.  Returns the running maximum of |v| (bits).
// off16[r]: byte offset of pixel r's line (32-channel group included); rows beyond the tensor lie beyond rs's range.
template <bool NT, int PITCH>
__device__ __forceinline__ unsigned store_p16_rows(const v16f& acc, float sc, float sh, int relu, float oscale, const __amdgpu_buffer_rsrc_t& rs,
                                                   unsigned base, int lane, unsigned tmax, unsigned rowmask = 0xffffu) {
    // base: byte offset of accumulator row 0's line (32-channel group included); row r lies (r & 3) + 8 (r >> 2) pixels further
    const int n = lane & 31;
    const bool even = (n & 1) == 0;
    base += even ? 2u * (unsigned)n : 64u + 2u * (unsigned)(n - 1);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        float v0 = fmaf(acc[r], sc, sh), v1 = fmaf(acc[r + 1], sc, sh);
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        // (rowmask: rows that exist - a ragged last tile's other rows are dropped by the store but must not reach the maximum)
        const unsigned a0 = ((rowmask >> r) & 1u) ? (__builtin_bit_cast(unsigned, v0) & 0x7fffffffu) : 0u;
        const unsigned a1 = ((rowmask >> (r + 1)) & 1u) ? (__builtin_bit_cast(unsigned, v1) & 0x7fffffffu) : 0u;
        tmax = a0 > tmax ? a0 : tmax;
        tmax = a1 > tmax ? a1 : tmax;
        unsigned h2, l2;  // (hi_r | hi_r+1 << 16), (lo_r | lo_r+1 << 16)
        f16_split2(v0 * oscale, v1 * oscale, h2, l2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned mine = k == 0 ? ((h2 & 0xffffu) | (l2 << 16)) : ((h2 >> 16) | (l2 & 0xffff0000u));  // (hi | lo << 16) of row r + k
            const unsigned nbr = (unsigned)__shfl_xor((int)mine, 1, 64);
            const unsigned outw = even ? ((mine & 0xffffu) | (nbr << 16)) : ((nbr >> 16) | (mine & 0xffff0000u));
            const unsigned off = base + (unsigned)((((r + k) & 3) + 8 * ((r + k) >> 2)) * PITCH);
            if (NT) __builtin_amdgcn_raw_buffer_store_b32(outw, rs, off, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b32(outw, rs, off, 0, 0);
        }
    }
    return tmax;
}

struct HaloParams {
    const char* x;        // P16 NHWC [B][H][W][CIN]
    const char* w;        // P16 [COUT][9 * CIN] (k = tap * CIN + c)
    float* y;             // fp32 [B][H][W][COUT]
    float* stats;         // [B * H / TH][COUT][4] = (mean, M2, min, max) per step tile of TH * W pixels, or null
    const float* x_amax;
    const float* w_amax;
    int B, H, W;
    int TH;               // image rows per step
    int R;                // ring rows = 2 * TH + 2
    int rowb;             // ring row pitch in bytes (a multiple of 1024 >= (W + 2) * pixel bytes)
    int rg_per_chunk;     // steps per chunk; a chunk = a band of rows of ONE image
    int chunks_per_image;
    int nchunks;
    FastDiv fdW;
    FastDiv fd_cpr;       // DMA chunk -> ring row
    int nt;               // non-temporal image loads / output stores (tensors far beyond the Infinity Cache)
    // eval mode (out16 != null; y / stats unused): out16 = act(y * bn_scale[c] + bn_shift[c]) as a P16 tensor [B][H][W][COUT]
    // scaled by the analytic bound of gemm_common.h EvalBound; the true maximum is folded into *ev.out_tmax
    char* out16;
    const float* bn_scale;
    const float* bn_shift;
    int relu;
    int pool;             // eval mode: the output is the 2x2 average of act(.) ([B][H/2][W/2][COUT]); 32 -> 64 channels, W % 64 == 0
    EvalBound ev;
};

// 16-byte unit u of pixel slot q is stored at unit u ^ swz(q) of that pixel's LDS line(s): the 16-lane groups of a
// ds_read_b128 fragment read (lanes = consecutive pixels, one unit) then cover all 64 banks for ANY pixel offset
template <int PIXB>
__device__ __forceinline__ int swz(int q) {
    return PIXB == 128 ? ((q >> 1) & 7) : (q & 15);
}

}  // namespace

// CIN, COUT in {32, 64}.  8 waves = PB pixel blocks x (COUT / 32) column blocks x (CIN / 32) channel groups;
// a wave multiplies its 32 pixels x 32 input channels x 9 taps into 32 output channels.
// EVAL: 1 = the eval-mode epilogue (its own instantiation: the training kernel keeps its register budget); 2 = ... followed by
// the 2x2 average pool (the stem's conv3, m_resnet.py:205-207: avgpool(relu(bn3(conv3(x))))), written pooled: pixel pairs along x
// are neighbouring accumulator rows of a lane, the two image rows of a pair are two waves that meet in LDS once per step
template <int CIN, int COUT, int EVAL = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_p16_kernel(HaloParams p) {
    constexpr int NW = 8, CB = COUT / 32, KG = CIN / 32, PB = NW / (CB * KG);
    constexpr int PIXB = CIN * 4;           // bytes of one pixel in LDS / HBM
    constexpr int UPP = PIXB / 16;          // 16-byte units (= loader lanes) per pixel
    constexpr int PPC = 64 / UPP;           // pixels per 1-KB DMA chunk
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    char* const ring = reinterpret_cast<char*>(smem);
    float4* const sstat = reinterpret_cast<float4*>(ring + (size_t)p.R * p.rowb);  // [2][NW][32]
    float* const ksum = reinterpret_cast<float*>(sstat + 2 * NW * 32);              // KG == 2: [PB * CB][16][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = wave / (CB * KG), cb = (wave / KG) % CB, kg = wave % KG;
    const int khalf = lane >> 5;
    const int W = p.W, H = p.H, TH = p.TH, R = p.R, rowb = p.rowb;
    const int cpr = rowb >> 10;  // DMA chunks per ring row

    const float unscale = 1.f / (f16_scale_of(*p.x_amax) * f16_scale_of(*p.w_amax));
    // eval mode: this lane's channel coefficients, the output scale from the analytic bound
    float e_sc = 1.f, e_sh = 0.f, e_oscale = 1.f;
    unsigned e_tmax = 0;
    if constexpr (EVAL) {
        e_sc = p.bn_scale[cb * 32 + (lane & 31)];
        e_sh = p.bn_shift[cb * 32 + (lane & 31)];
        const float bound = eval_out_bound(p.ev);
        if (p.ev.out_bound != nullptr && blockIdx.x == 0 && tid == 0) *p.ev.out_bound = bound;
        e_oscale = f16_scale_of(bound);
    }

    // ---- this wave's filter slice: B fragments [tap][k step][plane], 144 VGPRs, loaded once
    f16x8 bf[9][2][2];
    {
        const char* wr = p.w + (size_t)(cb * 32 + (lane & 31)) * (9 * CIN * 4);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    bf[t][ks][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wr + (t * KG + kg) * 128 + (4 * pl + 2 * ks + khalf) * 16));
    }

    // ---- this lane's pixel of the step tile (row-major over TH x W) and its three shifted LDS columns
    const int pp = pb * 32 + (lane & 31);
    const int ty = (int)fdiv((uint32_t)pp, p.fdW);
    const int tx = pp - ty * W;
    int qoff[3], qsw[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int q = tx + d;  // pixel slot of (x + d - 1): slot 0 is the zero column left of the image
        qoff[d] = q * PIXB;
        qsw[d] = swz<PIXB>(q);
    }

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((size_t)p.B * H * W * PIXB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(EVAL ? (void*)p.out16 : (void*)p.y, 0, (unsigned)((size_t)p.B * H * W * COUT * 4 / (EVAL == 2 ? 4 : 1)), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    static_assert(EVAL != 2 || KG == 1, "the pooled epilogue is built for 32 input channels (every wave holds final values)");
    // EVAL == 2: this wave's image row inside the step (pixel blocks are 32 pixels: W / 32 of them per row) and its partner
    const int wpb = W >> 5;
    const int prow = EVAL == 2 ? pb / wpb : 0, pbx = EVAL == 2 ? pb - prow * wpb : 0;
    const bool pool_owner = (prow & 1) == 0;  // even rows write the pooled pixels, odd rows hand their half over

    // rows [y_first, y_first + nrows) of image b -> ring slots slot_first ... (mod R); rows outside the image and the
    // pixel slots outside [1, W] become zeros (an out-of-range DMA lane writes zeros)
    auto issue_rows = [&](int b, int y_first, int nrows, int slot_first) {
        const int total = nrows * cpr;
        for (int i = wave; i < total; i += NW) {
            const int rr = (int)fdiv((uint32_t)i, p.fd_cpr), c = i - rr * cpr;
            const int y = y_first + rr;
            int slot = slot_first + rr;
            slot = slot >= R ? slot - R : slot;
            const int q = c * PPC + lane / UPP;  // pixel slot
            const int j = lane % UPP;            // stored unit position
            const int u = j ^ swz<PIXB>(q);      // source unit
            const int x = q - 1;
            const bool ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
            const unsigned vo = ok ? (unsigned)((((size_t)b * H + y) * W + x) * PIXB + u * 16) : OOB;
            dma16(rsX, ring + (size_t)slot * rowb + c * 1024, vo, p.nt != 0);
        }
    };

    for (int chunk = blockIdx.x; chunk < p.nchunks; chunk += gridDim.x) {
        const int b = chunk / p.chunks_per_image;
        const int rg0 = (chunk - b * p.chunks_per_image) * p.rg_per_chunk;
        int y0 = rg0 * TH;
        int slot0 = 0;  // ring slot of image row y0 - 1
        issue_rows(b, y0 - 1, TH + 2, 0);
        for (int s = 0; s < p.rg_per_chunk; ++s) {
            // rows y0-1 .. y0+TH were issued before this wave's stores of the previous step (vmcnt retires in order)
            if (s == 0) wait_vm<0>();
            else if (EVAL == 2) { if (pool_owner) wait_vm<8>(); else wait_vm<0>(); }  // (8 pooled stores per lane, or none)
            else if (KG == 1 || kg == 0) wait_vm<16>();
            else wait_vm<0>();
            lds_barrier();
            // previous step's BatchNorm partials: merged by the first lanes of wave 0 (one column each)
            if (!EVAL && p.stats != nullptr && s > 0 && wave == 0 && lane < COUT) {
                const float4* src = sstat + ((s - 1) & 1) * NW * 32;
                const int c2 = lane >> 5, n = lane & 31;
                float cnt = 0.f, mean = 0.f, m2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const float4 v = src[((k * CB + c2) * KG) * 32 + n];
                    const float nt = cnt + 32.f, d = v.x - mean;
                    mean += d * (32.f / nt);
                    m2 += v.y + d * d * (cnt * 32.f / nt);
                    cnt = nt;
                    lo = fminf(lo, v.z);
                    hi = fmaxf(hi, v.w);
                }
                const long long part = ((long long)b * H + (y0 - TH)) / TH;
                reinterpret_cast<float4*>(p.stats)[part * COUT + lane] = make_float4(mean, m2, lo, hi);
                // (no wait here: this store is OLDER than the row DMA issued next, and vmcnt retires in order - the counted wait at
                // the top of the next step, which leaves only the 16 output stores issued after that DMA in flight, covers it.
                // Draining it on the spot cost wave 0 a memory round trip per step while seven waves waited at the barrier)
            }
            if (s + 1 < p.rg_per_chunk) {
                int sl = slot0 + TH + 2;
                sl = sl >= R ? sl - R : sl;
                issue_rows(b, y0 + TH + 1, TH, sl);
            }

            // ---- 9 taps x 2 k steps x 3 products on the TH + 2 ring rows; the fragments of group g + 1 are fetched
            // before the MFMAs of group g issue (two register sets, statically indexed)
            v16f acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const char* rowp[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                int rs = slot0 + ty + dy;
                rs = rs >= R ? rs - R : rs;
                rs = rs >= R ? rs - R : rs;
                rowp[dy] = ring + (size_t)rs * rowb;
            }
            f16x8 af[2][2];
            auto fetch = [&](int g, f16x8(&dst)[2]) {  // g = tap * 2 + ks
                const int t = g >> 1, ks = g & 1, dy = t / 3, dx = t - 3 * dy;
                const char* px = rowp[dy] + qoff[dx];
                const int u0 = (KG == 2 ? kg * 8 : 0) + 2 * ks + khalf;
                dst[0] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(px + (((u0) ^ qsw[dx]) << 4)));
                dst[1] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(px + (((u0 + 4) ^ qsw[dx]) << 4)));
            };
            fetch(0, af[0]);
#pragma unroll
            for (int g = 0; g < 18; ++g) {
                if (g + 1 < 18) fetch(g + 1, af[(g + 1) & 1]);
                const int t = g >> 1, ks = g & 1;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[g & 1][0], bf[t][ks][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[g & 1][1], bf[t][ks][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[g & 1][0], bf[t][ks][0], acc, 0, 0, 0);
            }

            // ---- epilogue.  The scale (a power of two: exact) comes off first, in every wave: the accumulators then reach
            // the inline-asm LDS stores below through a VALU instruction - straight from the last MFMA the compiler does
            // not know the asm is an LDS read of the MFMA's destination and leaves out the wait states that hazard needs
            // (element 0 of the <64, 64> form was stored before its final value had landed)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= unscale;
            if constexpr (KG == 2) {  // the two channel-group halves of a pixel block meet in LDS
                float* dst = ksum + (pb * CB + cb) * (16 * 64);
                if (kg == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[r];
                        lds_store1(dst + r * 64 + lane, v);
                    }
                }
                lds_barrier();
                if (kg == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += dst[r * 64 + lane];
                }
            }
            if (KG == 1 || kg == 0) {
                if (!EVAL && p.stats != nullptr) {
                    float sum = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sum += acc[r];
                        lo = fminf(lo, acc[r]);
                        hi = fmaxf(hi, acc[r]);
                    }
                    sum += __shfl_xor(sum, 32, 64);
                    const float mean = sum * (1.f / 32.f);
                    float m2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[r] - mean;
                        m2 += d * d;
                    }
                    m2 += __shfl_xor(m2, 32, 64);
                    lo = fminf(lo, __shfl_xor(lo, 32, 64));
                    hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
                    if (khalf == 0) lds_store4(sstat + (s & 1) * NW * 32 + wave * 32 + (lane & 31), make_float4(mean, m2, lo, hi));
                }
                const size_t m0 = ((size_t)b * H + y0) * W + pb * 32 + 4 * khalf;
                if constexpr (EVAL == 2) {
                    // BatchNorm + ReLU, then the 2x2 average: along x in the lane (accumulator rows 2j, 2j + 1 are neighbouring
                    // pixels), along y through LDS (the odd image rows' waves hand their sums to the even rows' waves)
                    float hs[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v0 = fmaf(acc[2 * j], e_sc, e_sh), v1 = fmaf(acc[2 * j + 1], e_sc, e_sh);
                        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        hs[j] = v0 + v1;
                    }
                    float* ex = reinterpret_cast<float*>(sstat);  // [NW][8][64] floats = 16 KB ... of which the odd rows' waves write
                    if (!pool_owner) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float v = hs[j];
                            lds_store1(ex + (wave * 8 + j) * 64 + lane, v);
                        }
                    }
                    lds_barrier();
                    if (pool_owner) {
                        const int partner = wave + wpb * CB * KG;  // the wave one image row below, same columns / channels
                        v16f pv;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pv[j] = 0.25f * (hs[j] + ex[(partner * 8 + j) * 64 + lane]);
                        // pooled pixel of pair j: block of 16 pooled pixels per 32-pixel block, index 2 khalf + (j & 1) + 4 (j >> 1)
                        const size_t pm0 = ((size_t)b * (H >> 1) + ((y0 + prow) >> 1)) * (W >> 1) + pbx * 16 + 2 * khalf;
                        const unsigned base = (unsigned)(pm0 * (COUT * 4)) + (unsigned)cb * 128u;
                        const int n = lane & 31;
                        const bool even = (n & 1) == 0;
                        const unsigned colb = base + (even ? 2u * (unsigned)n : 64u + 2u * (unsigned)(n - 1));
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const float v0 = pv[j], v1 = pv[j + 1];
                            const unsigned a0 = __builtin_bit_cast(unsigned, v0) & 0x7fffffffu, a1 = __builtin_bit_cast(unsigned, v1) & 0x7fffffffu;
                            e_tmax = a0 > e_tmax ? a0 : e_tmax;
                            e_tmax = a1 > e_tmax ? a1 : e_tmax;
                            unsigned h2, l2;
                            f16_split2(v0 * e_oscale, v1 * e_oscale, h2, l2);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const unsigned mine = k == 0 ? ((h2 & 0xffffu) | (l2 << 16)) : ((h2 >> 16) | (l2 & 0xffff0000u));
                                const unsigned nbr = (unsigned)__shfl_xor((int)mine, 1, 64);
                                const unsigned outw = even ? ((mine & 0xffffu) | (nbr << 16)) : ((nbr >> 16) | (mine & 0xffff0000u));
                                const unsigned off = colb + (unsigned)((((j + k) & 1) + 4 * ((j + k) >> 1)) * (COUT * 4));
                                __builtin_amdgcn_raw_buffer_store_b32(outw, rsY, off, 0, 0);
                            }
                        }
                    }
                } else if constexpr (EVAL == 1) {  // BatchNorm (running statistics) + ReLU, written as the next operand (16 stores per lane as well)
                    const unsigned base = (unsigned)(m0 * (COUT * 4)) + (unsigned)cb * 128u;
                    e_tmax = p.nt ? store_p16_rows<true, COUT * 4>(acc, e_sc, e_sh, p.relu, e_oscale, rsY, base, lane, e_tmax)
                                  : store_p16_rows<false, COUT * 4>(acc, e_sc, e_sh, p.relu, e_oscale, rsY, base, lane, e_tmax);
                } else {
                    const unsigned col = (unsigned)(cb * 32 + (lane & 31)) * 4u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned off = (unsigned)((m0 + (r & 3) + 8 * (r >> 2)) * (COUT * 4)) + col;
                        const float v = acc[r];  // (a bit_cast of the vector ELEMENT itself stores element 0 sixteen times)
                        if (p.nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, off, 0, 2);
                        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, off, 0, 0);
                    }
                }
            }
            y0 += TH;
            slot0 += TH;
            slot0 = slot0 >= R ? slot0 - R : slot0;
        }
        // chunk end: the last step's partials, and every wave is done with the ring before the next chunk primes it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (!EVAL && p.stats != nullptr && wave == 0 && lane < COUT) {
            const int s = p.rg_per_chunk;
            const float4* src = sstat + ((s - 1) & 1) * NW * 32;
            const int c2 = lane >> 5, n = lane & 31;
            float cnt = 0.f, mean = 0.f, m2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int k = 0; k < PB; ++k) {
                const float4 v = src[((k * CB + c2) * KG) * 32 + n];
                const float nt = cnt + 32.f, d = v.x - mean;
                mean += d * (32.f / nt);
                m2 += v.y + d * d * (cnt * 32.f / nt);
                cnt = nt;
                lo = fminf(lo, v.z);
                hi = fmaxf(hi, v.w);
            }
            const long long part = ((long long)b * H + (y0 - TH)) / TH;
            reinterpret_cast<float4*>(p.stats)[part * COUT + lane] = make_float4(mean, m2, lo, hi);
        }
        lds_barrier();
    }
    if (EVAL && p.ev.out_tmax != nullptr) {  // one atomic per (persistent) workgroup
        e_tmax = wave_umax(e_tmax);
        unsigned* redu = reinterpret_cast<unsigned*>(sstat);
        if (lane == 0) redu[wave] = e_tmax;
        lds_barrier();
        if (tid == 0) {
            unsigned r = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) r = redu[w] > r ? redu[w] : r;
            if (r != 0) atomicMax(reinterpret_cast<unsigned*>(p.ev.out_tmax), r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- conv1
// y[b, yo, xo, n] = sum_{c, ky, kx} img[b, c, 2 yo - 1 + ky, 2 xo - 1 + kx] * w[n, c, ky, kx]   (3x3, stride 2, pad 1)
// on v_mfma_f32_32x32x2_f32 (exact fp32): a wave = 32 consecutive output pixels x all 32 output channels, K = 27 (+1 zero)
// in 14 steps; A[i = pixel][k] gathered straight from the NCHW image (out-of-range taps read as zero through the buffer
// descriptor), B[k][n] = the filter, 14 VGPRs for the whole launch.  A workgroup = 4 waves = one 128-row BatchNorm slab:
// (mean, M2, min, max) per channel, Chan-merged over its four 32-row wave partials.
struct Conv1Params {
    const float* img;   // [B][3][Hi][Wi]
    const float* w;     // [32][27]
    float* y;           // [B][Ho][Wo][32]
    float* stats;       // [ceil(M / 128)][32][4] or null
    int B, Hi, Wi, Ho, Wo;
    long long M;        // B * Ho * Wo
    int nslabs;
    FastDiv fdWo, fdHo;
    // eval mode (out16 != null; y / stats unused): out16 = act(y * bn_scale + bn_shift) as a P16 tensor [B][Ho][Wo][32]
    char* out16;
    const float* bn_scale;
    const float* bn_shift;
    int relu;
    EvalBound ev;
};

template <bool EVAL>
__global__ __launch_bounds__(256) void stem_conv1_kernel(Conv1Params p) {
    __shared__ float4 sstat[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, n = lane & 31;
    float bw[14];
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) {
        const int j = 2 * kk + kh;
        bw[kk] = j < 27 ? p.w[n * 27 + j] : 0.f;
    }
    const size_t img_bytes = (size_t)p.B * 3 * p.Hi * p.Wi * 4;
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, (unsigned)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(EVAL ? (void*)p.out16 : (void*)p.y, 0, (unsigned)(p.M * 128), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    float e_sc = 1.f, e_sh = 0.f, e_oscale = 1.f;
    unsigned e_tmax = 0;
    if constexpr (EVAL) {
        e_sc = p.bn_scale[n];
        e_sh = p.bn_shift[n];
        const float bound = eval_out_bound(p.ev);
        if (p.ev.out_bound != nullptr && blockIdx.x == 0 && tid == 0) *p.ev.out_bound = bound;
        e_oscale = f16_scale_of(bound);
    }
    int it = 0;
    for (int slab = blockIdx.x; slab < p.nslabs; slab += gridDim.x, ++it) {
        const long long m = (long long)slab * 128 + wave * 32 + (lane & 31);
        const bool live = m < p.M;
        const uint32_t mm = live ? (uint32_t)m : 0u;
        const uint32_t q = fdiv(mm, p.fdWo);
        const int xo = (int)(mm - q * p.Wo);
        const uint32_t b = fdiv(q, p.fdHo);
        const int yo = (int)(q - b * p.Ho);
        // branch-free gather: the byte offset of every tap is formed in 32-bit arithmetic whether or not the tap exists,
        // a tap outside the image (or k = 27, or a pixel beyond M) gets the out-of-range offset and reads as zero
        const unsigned img0 = (unsigned)b * 3u * (unsigned)(p.Hi * p.Wi);
        float a[14];
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) {
            const int j0 = 2 * kk, j1 = 2 * kk + 1;  // this lane's k index is j0 (lower half-wave) or j1 (upper)
            const int c = kh ? j1 / 9 : j0 / 9, r9 = kh ? j1 % 9 : j0 % 9;
            const int ky = kh ? (j1 % 9) / 3 : (j0 % 9) / 3, kx = r9 - ky * 3;
            const int yy = 2 * yo - 1 + ky, xx = 2 * xo - 1 + kx;
            const bool ok = live & ((2 * kk + kh) < 27) & (yy >= 0) & (yy < p.Hi) & (xx >= 0) & (xx < p.Wi);
            const unsigned off = (img0 + (unsigned)((c * p.Hi + yy) * p.Wi + xx)) * 4u;
            a[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsI, ok ? off : OOB, 0, 0));
        }
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bw[kk], acc, 0, 0, 0);
        // rows of this wave that exist
        const long long row0 = (long long)slab * 128 + wave * 32;
        const int cnt_w = (int)(p.M - row0 < 32 ? (p.M - row0 > 0 ? p.M - row0 : 0) : 32);
        if (!EVAL && p.stats != nullptr) {
            float sum = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < cnt_w) {
                    sum += acc[r];
                    lo = fminf(lo, acc[r]);
                    hi = fmaxf(hi, acc[r]);
                }
            }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = cnt_w > 0 ? sum / (float)cnt_w : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float d = acc[r] - mean;
                if (row < cnt_w) m2 += d * d;
            }
            m2 += __shfl_xor(m2, 32, 64);
            lo = fminf(lo, __shfl_xor(lo, 32, 64));
            hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
            if (kh == 0) sstat[it & 1][wave][n] = make_float4(mean, m2, lo, hi);
        }
        if constexpr (EVAL) {
            // rows beyond M: their stores are dropped (descriptor range), and they are kept out of the maximum
            unsigned rowmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) rowmask |= (((r & 3) + 8 * (r >> 2) + 4 * kh) < cnt_w ? 1u : 0u) << r;
            e_tmax = store_p16_rows<false, 128>(acc, e_sc, e_sh, p.relu, e_oscale, rsY, (unsigned)((row0 + 4 * kh) * 128), lane, e_tmax, rowmask);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // (rows beyond M lie beyond the descriptor's range: the store is dropped)
                const unsigned off = (unsigned)((row0 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 128 + n * 4);
                const float v = acc[r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, off, 0, 0);
            }
        }
        if (!EVAL && p.stats != nullptr) {
            __syncthreads();  // (double-buffered: the next slab's partials go to the other half)
            if (wave == 0 && lane < 32) {
                float cnt = 0.f, mean = 0.f, m2 = 0.f, lo = INFINITY, hi = -INFINITY;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long long r0 = (long long)slab * 128 + k * 32;
                    const float nb = (float)(p.M - r0 < 32 ? (p.M - r0 > 0 ? p.M - r0 : 0) : 32);
                    if (nb > 0.f) {
                        const float4 v = sstat[it & 1][k][lane];
                        const float nt = cnt + nb, d = v.x - mean;
                        mean += d * (nb / nt);
                        m2 += v.y + d * d * (cnt * nb / nt);
                        cnt = nt;
                        lo = fminf(lo, v.z);
                        hi = fmaxf(hi, v.w);
                    }
                }
                reinterpret_cast<float4*>(p.stats)[(long long)slab * 32 + lane] = make_float4(mean, m2, lo, hi);
            }
        }
    }
    if (EVAL && p.ev.out_tmax != nullptr) {  // one atomic per workgroup (at most 2048 of them)
        e_tmax = wave_umax(e_tmax);
        __syncthreads();
        unsigned* redu = reinterpret_cast<unsigned*>(&sstat[0][0][0]);
        if (lane == 0) redu[wave] = e_tmax;
        __syncthreads();
        if (tid == 0) {
            const unsigned a = redu[0] > redu[1] ? redu[0] : redu[1], b = redu[2] > redu[3] ? redu[2] : redu[3];
            const unsigned r = a > b ? a : b;
            if (r != 0) atomicMax(reinterpret_cast<unsigned*>(p.ev.out_tmax), r);
        }
    }
}

// conv1's weight gradient, dW[n][j] = sum over output pixels m of dy[m][n] * patch[m][j] (j = c * 9 + ky * 3 + kx, the image
// value under tap j of pixel m), again straight from the NCHW image: on v_mfma_f32_32x32x2_f32 the OUTPUT is the [32][27 (+5)]
// gradient itself and the reduction runs over pixels, two per MFMA - lane (col, k) feeds A with dy[pixel k][channel col]
// (a coalesced 128-byte row per half-wave) and B with the image value under ITS tap col of pixel k (out-of-range taps, the
// five unused columns and pixels beyond M read as zero through the buffer descriptor).  Every wave owns a contiguous run of
// pixels (16 loads in flight per lane), the four waves of a workgroup meet in LDS, the workgroups' [32][27] slabs are folded
// by trid_slab_reduce_f32: a fixed assignment and order, so the result is reproducible.
struct Conv1WgradParams {
    const float* img;   // [B][3][Hi][Wi]
    const float* dy;    // [B][Ho][Wo][32]
    float* slabs;       // [gridDim.x][32 * 27]
    int B, Hi, Wi, Ho, Wo;
    long long M;        // B * Ho * Wo
    int per_wave;       // pixels per wave (a multiple of 16)
    FastDiv fdWo, fdHo;
};

constexpr int C1W_SLABS = 512;  // workgroups (= slabs) of the weight-gradient launch

__global__ __launch_bounds__(256) void stem_conv1_wgrad_kernel(Conv1WgradParams p) {
    __shared__ float red[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, j = lane & 31;
    const int c = j / 9, ky = (j % 9) / 3, kx = j % 3;
    const bool tap_live = j < 27;
    const int tap_off = (c * p.Hi + ky - 1) * p.Wi + (kx - 1);  // relative to pixel (2 yo, 2 xo) of plane 0
    const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, (unsigned)((size_t)p.B * 3 * p.Hi * p.Wi * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)(p.M * 128), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    constexpr int U = 8;
    const long long first = ((long long)blockIdx.x * 4 + wave) * p.per_wave;
    long long last = first + p.per_wave;
    last = last < p.M ? last : p.M;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (long long t = first; t < last; t += 2 * U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = t + 2 * u + kh;
            const bool live = m < last;
            const uint32_t mm = live ? (uint32_t)m : 0u;
            a[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, live ? mm * 128u + (unsigned)j * 4u : OOB, 0, 0));
            const uint32_t q = fdiv(mm, p.fdWo);
            const int xo = (int)(mm - q * p.Wo);
            const uint32_t bi = fdiv(q, p.fdHo);
            const int yo = (int)(q - bi * p.Ho);
            const int yy = 2 * yo - 1 + ky, xx = 2 * xo - 1 + kx;
            const bool ok = live & tap_live & (yy >= 0) & (yy < p.Hi) & (xx >= 0) & (xx < p.Wi);
            const unsigned off = (bi * 3u * (unsigned)(p.Hi * p.Wi) + (unsigned)(2 * yo * p.Wi + 2 * xo + tap_off)) * 4u;
            b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsI, ok ? off : OOB, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kh][j] = acc[r];
    __syncthreads();
    float* out = p.slabs + (size_t)blockIdx.x * (32 * 27);
    for (int e = tid; e < 32 * 27; e += 256) {
        const int n = e / 27, jj = e - n * 27;
        out[e] = (red[0][n][jj] + red[1][n][jj]) + (red[2][n][jj] + red[3][n][jj]);
    }
}

// ---------------------------------------------------------------------------------------------------- 3x3 weight gradients
// dW[n][tap][c] = sum over pixels of dy[pixel][n] * x[pixel + tap offset][c] for the same layers (32 / 64 channels, the
// largest maps).  The transposing GEMM kernel (gemm_p16.hip) stages the shifted image once per TAP: nine passes of x through
// the L2 -> LDS path (3.5 GB per launch for the stem's conv2, 15 TB/s: its limit).  Here the rings of the forward kernel
// return: TH + 2 rows of x (with the zero padding as data) and TH rows of dy in LDS, each pixel staged ONCE, and the nine
// taps are nine shifted TRANSPOSING fragment reads (ds_read_b64_tr_b16: a lane gets 4 consecutive pixels of its channel)
// of the same x image against one dy fragment.  The reduction runs over pixels (16 per MFMA), the OUTPUT tile of a wave -
// 32 dy channels x 32 x channels x 9 taps - lives in 144 accumulator VGPRs for the whole launch; waves split the pixels
// of a step, meet in LDS at the end, and every workgroup writes one [N][9 C] slab (folded by trid_slab_reduce_f32 in a fixed
// order).  LDS row image: [32-channel group][pixel slot][128 B], 16-byte slot s of pixel slot q stored at
// s ^ (4 * ((q >> 1) & 1)) - the reader applies the swizzle of the slot it actually reads, so any tap shift stays
// conflict-free (rows q and q + 2 of a 4-pixel tile share banks and differ in that bit).
struct HaloWgradParams {
    const char* x;        // P16 NHWC [B][H][W][CIN]
    const char* dy;       // P16 NHWC [B][H][W][COUT]
    float* slabs;         // [gridDim.x][COUT][9 * CIN]
    const float* x_amax;
    const float* dy_amax;
    int B, H, W, TH;
    int xcpg;             // 1-KB DMA chunks per channel group of an x ring row: ceil((W + 2) / 8)
    int rg_per_chunk, chunks_per_image, nchunks;
    FastDiv fd_xrow, fd_xcpg, fd_drow, fd_dcpg, fdW;  // chunk -> (row, group, pixel block); pixel -> (row, column)
};

typedef __fp16 h4v_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ f16x8 tr_frag8(const char* lds_addr) {  // 8 pixels of this lane's channel: two 4-pixel transposing reads
    typedef __attribute__((address_space(3))) h4v_t* lp;
    const h4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_addr));
    const h4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lp)(lds_addr + 512));
    f16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// 8 waves = (COUT / 32 dy-channel blocks) x (CIN / 32 x-channel blocks) x PG pixel groups
template <int CIN, int COUT>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_halo_p16_kernel(HaloWgradParams p) {
    constexpr int NW = 8, NB = COUT / 32, CBK = CIN / 32, PG = NW / (NB * CBK);
    extern __shared__ __attribute__((aligned(16))) uint4 smem[];
    const int W = p.W, H = p.H, TH = p.TH;
    const int RX = 2 * TH + 2;
    const int xgs = p.xcpg * 1024, xrowb = CBK * xgs;  // x ring: bytes per channel-group plane / per row
    const int dgs = W * 128, drowb = NB * dgs;          // dy ring
    char* const xring = reinterpret_cast<char*>(smem);
    char* const dring = xring + (size_t)RX * xrowb;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wave / (CBK * PG), cbk = (wave / PG) % CBK, pg = wave % PG;
    // transposing-read lane constants: 16-channel half h, pixel r4 of the 4 x 16 tile, 8-byte chunk c8, k half kh
    const int h = (lane >> 4) & 1, r4 = (lane & 15) >> 2, c8 = lane & 3, kh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)((size_t)p.B * H * W * CIN * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)((size_t)p.B * H * W * COUT * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // x rows [y_first, y_first + nrows) of image b -> ring slots slot_first .. (mod RX); slot q of a row is image column q - 1
    auto issue_x = [&](int b, int y_first, int nrows, int slot_first) {
        const int per_row = CBK * p.xcpg;
        for (int i = wave; i < nrows * per_row; i += NW) {
            const int rr = (int)fdiv((uint32_t)i, p.fd_xrow), c = i - rr * per_row;
            const int g = CBK == 1 ? 0 : (int)fdiv((uint32_t)c, p.fd_xcpg), q = 8 * (c - g * p.xcpg) + (lane >> 3);
            int slot = slot_first + rr;
            slot = slot >= RX ? slot - RX : slot;
            const int y = y_first + rr, xx = q - 1;
            const int su = (lane & 7) ^ (((q >> 1) & 1) << 2);
            const bool ok = (y >= 0) && (y < H) && (xx >= 0) && (xx < W);
            const unsigned vo = ok ? (unsigned)((((size_t)b * H + y) * W + xx) * (CIN * 4) + g * 128 + su * 16) : OOB;
            dma16(rsX, xring + (size_t)slot * xrowb + c * 1024, vo, false);
        }
    };
    // dy rows [y_first, y_first + TH) -> ring rows half * TH ..
    auto issue_d = [&](int b, int y_first, int half) {
        const int cpg = W / 8, per_row = NB * cpg;
        for (int i = wave; i < TH * per_row; i += NW) {
            const int rr = (int)fdiv((uint32_t)i, p.fd_drow), c = i - rr * per_row;
            const int g = NB == 1 ? 0 : (int)fdiv((uint32_t)c, p.fd_dcpg), q = 8 * (c - g * cpg) + (lane >> 3);
            const int su = (lane & 7) ^ (((q >> 1) & 1) << 2);
            const unsigned vo = (unsigned)((((size_t)b * H + (y_first + rr)) * W + q) * (COUT * 4) + g * 128 + su * 16);
            dma16(rsD, dring + (size_t)(half * TH + rr) * drowb + c * 1024, vo, false);
        }
    };

    v16f acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ksteps = TH * W / 16;      // 16-pixel reduction steps per step tile
    const int kper = ksteps / PG;        // ... of this wave (the host guarantees divisibility)
    // byte offset of this lane's fragment piece inside a pixel's 128-byte line: slot (plane, half, chunk), swizzled by the pixel slot
    auto piece = [&](int q, int pl) { return q * 128 + ((((pl * 4 + h * 2 + (c8 >> 1))) ^ (((q >> 1) & 1) << 2)) << 4) + 8 * (c8 & 1); };

    for (int chunk = blockIdx.x; chunk < p.nchunks; chunk += gridDim.x) {
        const int b = chunk / p.chunks_per_image;
        int y0 = (chunk - b * p.chunks_per_image) * p.rg_per_chunk * TH;
        int slot0 = 0;  // ring slot of image row y0 - 1
        issue_x(b, y0 - 1, TH + 2, 0);
        issue_d(b, y0, 0);
        for (int s = 0; s < p.rg_per_chunk; ++s) {
            wait_vm<0>();  // (no stores in this loop: everything this wave has in flight are the rows of this step)
            lds_barrier();
            if (s + 1 < p.rg_per_chunk) {
                int sl = slot0 + TH + 2;
                sl = sl >= RX ? sl - RX : sl;
                issue_x(b, y0 + TH + 1, TH, sl);
                issue_d(b, y0 + TH, (s + 1) & 1);
            }
            const char* dbase = dring + (size_t)((s & 1) * TH) * drowb + nb * dgs;
            for (int kk = 0; kk < kper; ++kk) {
                const int pix = (pg * kper + kk) * 16;       // first pixel of this k step inside the step tile
                const int ty = (int)fdiv((uint32_t)pix, p.fdW), x0 = pix - ty * W;
                const int qd = x0 + 8 * kh + r4;
                const f16x8 ah = tr_frag8(dbase + (size_t)ty * drowb + piece(qd, 0));
                const f16x8 al = tr_frag8(dbase + (size_t)ty * drowb + piece(qd, 1));
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int rs = slot0 + ty + ky;
                    rs = rs >= RX ? rs - RX : rs;
                    rs = rs >= RX ? rs - RX : rs;
                    const char* xb = xring + (size_t)rs * xrowb + cbk * xgs;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int qx = x0 + kx + 8 * kh + r4;
                        const f16x8 bh = tr_frag8(xb + piece(qx, 0));
                        const f16x8 bl = tr_frag8(xb + piece(qx, 1));
                        const int t = ky * 3 + kx;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                    }
                }
            }
            y0 += TH;
            slot0 += TH;
            slot0 = slot0 >= RX ? slot0 - RX : slot0;
        }
        lds_barrier();  // every wave is done with the rings before the next chunk primes them
    }

    // ---- the pixel groups of a (dy block, x block) pair meet in LDS, tap by tap; one slab per workgroup
    const float f = 1.f / (f16_scale_of(*p.x_amax) * f16_scale_of(*p.dy_amax));
    float* const tile = reinterpret_cast<float*>(smem);  // [NW][32][33]
    float* const out = p.slabs + (size_t)blockIdx.x * (COUT * 9 * CIN);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + (lane & 31)] = f * acc[t][r];
        __syncthreads();
        for (int e = tid; e < NB * CBK * 1024; e += 512) {
            const int pair = e >> 10, i = (e >> 5) & 31, j = e & 31;  // pair = nb * CBK + cbk
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < PG; ++g) v += tile[((pair * PG + g) * 32 + i) * 33 + j];
            out[(size_t)((pair / CBK) * 32 + i) * (9 * CIN) + t * CIN + (pair % CBK) * 32 + j] = v;
        }
        __syncthreads();
    }
}

template <int CIN, int COUT, int EVAL>
static int launch_halo_t(HaloParams& p, hipStream_t stream) {
    constexpr int NW = 8, CB = COUT / 32, KG = CIN / 32, PB = NW / (CB * KG);
    // (the [2][NW][32] float4 partials region = 8 KB; the pooled epilogue uses [NW][8][64] floats = 16 KB of exchange space there)
    const size_t lds = (size_t)p.R * p.rowb + (EVAL == 2 ? 2 : 1) * 2 * NW * 32 * sizeof(float4) + (KG == 2 ? (size_t)PB * CB * 16 * 64 * 4 : 0);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)conv3x3_halo_p16_kernel<CIN, COUT, EVAL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_conv3x3_halo_p16: cannot reserve LDS: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    const int grid = std::min(p.nchunks, 256);
    hipLaunchKernelGGL((conv3x3_halo_p16_kernel<CIN, COUT, EVAL>), dim3(grid), dim3(512), lds, stream, p);
    return check_launch("trid_conv3x3_halo_p16");
}

template <int CIN, int COUT>
static int launch_halo(HaloParams& p, hipStream_t stream) {
    if constexpr (CIN == 32 && COUT == 64) {
        if (p.out16 != nullptr && p.pool) return launch_halo_t<CIN, COUT, 2>(p, stream);
    }
    return p.out16 != nullptr ? launch_halo_t<CIN, COUT, 1>(p, stream) : launch_halo_t<CIN, COUT, 0>(p, stream);
}

}  // namespace trid

using namespace trid;

// rows per step of the halo kernel for this geometry, or 0 when it does not apply (the caller then uses trid_gemm_p16)
static int halo_rows_per_step(int H, int W, int Cin, int Cout) {
    if (!((Cin == 32 || Cin == 64) && (Cout == 32 || Cout == 64))) return 0;
    const int pb = 8 / ((Cin / 32) * (Cout / 32));
    const int pix = pb * 32;
    if (W <= 0 || pix % W != 0) return 0;
    const int th = pix / W;
    if (H % th != 0) return 0;
    const int rowb = ((W + 2) * Cin * 4 + 1023) / 1024 * 1024;
    const size_t lds = (size_t)(2 * th + 2) * rowb + 2 * 8 * 32 * 16 + (Cin == 64 ? (size_t)pb * (Cout / 32) * 16 * 64 * 4 : 0);
    if (lds > 160 * 1024) return 0;
    return th;
}

extern "C" int trid_conv3x3_halo_rows(int H, int W, int Cin, int Cout) { return halo_rows_per_step(H, W, Cin, Cout); }

static int halo_dispatch(HaloParams& p, int B, int H, int W, int Cin, int Cout, int chunks_per_image, const char* who, hipStream_t s) {
    const int th = halo_rows_per_step(H, W, Cin, Cout);
    TRID_REQUIRE(th > 0, "%s: geometry H=%d W=%d Cin=%d Cout=%d is not covered (trid_conv3x3_halo_rows() == 0)", who, H, W, Cin, Cout);
    TRID_REQUIRE((long long)B * H * W * std::max(Cin, Cout) * 4 < (1ll << 31), "%s: tensors must stay below 2 GB (31-bit buffer offsets)", who);
    p.B = B; p.H = H; p.W = W;
    p.TH = th;
    p.R = 2 * th + 2;
    p.rowb = ((W + 2) * Cin * 4 + 1023) / 1024 * 1024;
    const int rg = H / th;
    int cpi = rg;  // chunks per image: the smallest divisor of the step count that gives the chip >= 256 chunks
    for (int d = 1; d <= rg; ++d)
        if (rg % d == 0 && (long long)B * d >= 256) {
            cpi = d;
            break;
        }
    if (chunks_per_image > 0) {
        TRID_REQUIRE(rg % chunks_per_image == 0, "%s: chunks_per_image must divide the %d steps of an image", who, rg);
        cpi = chunks_per_image;
    }
    p.chunks_per_image = cpi;
    p.rg_per_chunk = rg / cpi;
    p.nchunks = B * cpi;
    p.fdW = make_fastdiv((uint32_t)W);
    p.fd_cpr = make_fastdiv((uint32_t)(p.rowb >> 10));
    static const int nt_env = getenv("TRID_STREAM_NT") ? atoi(getenv("TRID_STREAM_NT")) : -1;
    p.nt = nt_env >= 0 ? (nt_env != 0) : ((long long)B * H * W * std::min(Cin, Cout) * 4 >= STREAM_NT_MIN_BYTES);
    if (Cin == 32 && Cout == 32) return launch_halo<32, 32>(p, s);
    if (Cin == 32 && Cout == 64) return launch_halo<32, 64>(p, s);
    if (Cin == 64 && Cout == 64) return launch_halo<64, 64>(p, s);  // (layer1's conv2, m_resnet.py:22)
    return launch_halo<64, 32>(p, s);
}

extern "C" int trid_conv3x3_halo_p16(const void* x, const float* x_amax, const void* w, const float* w_amax, float* y, float* stats,
                                     int B, int H, int W, int Cin, int Cout, int chunks_per_image, void* stream) {
    TRID_REQUIRE(x && w && y && x_amax && w_amax && B > 0 && H > 0 && W > 0, "trid_conv3x3_halo_p16: bad arguments");
    TRID_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && (!stats || aligned16(stats)), "trid_conv3x3_halo_p16: operands must be 16-byte aligned");
    HaloParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const char*)x; p.w = (const char*)w; p.y = y; p.stats = stats;
    p.x_amax = x_amax; p.w_amax = w_amax;
    return halo_dispatch(p, B, H, W, Cin, Cout, chunks_per_image, "trid_conv3x3_halo_p16", (hipStream_t)stream);
}

extern "C" int trid_conv3x3_halo_eval_pool_ok(int H, int W, int Cin, int Cout) {
    const int th = halo_rows_per_step(H, W, Cin, Cout);
    return Cin == 32 && Cout == 64 && th > 0 && th % 2 == 0 && W % 64 == 0 && H % 2 == 0;
}

extern "C" int trid_conv3x3_halo_eval_p16(const void* x, const float* x_amax, const void* w, const float* w_amax, const float* bn_scale,
                                          const float* bn_shift, void* out, const float* eval_coef, const float* eval_tin, float* out_bound,
                                          float* out_tmax, int B, int H, int W, int Cin, int Cout, int relu, int pool, void* stream) {
    TRID_REQUIRE(!pool || trid_conv3x3_halo_eval_pool_ok(H, W, Cin, Cout), "trid_conv3x3_halo_eval_p16: the pooled form covers 32 -> 64 channels, W %% 64 == 0 (H=%d W=%d Cin=%d Cout=%d)", H, W, Cin, Cout);
    TRID_REQUIRE(x && w && out && x_amax && w_amax && bn_scale && bn_shift && eval_coef && eval_tin && out_bound && B > 0 && H > 0 && W > 0,
                 "trid_conv3x3_halo_eval_p16: bad arguments");
    TRID_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out), "trid_conv3x3_halo_eval_p16: operands must be 16-byte aligned");
    HaloParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const char*)x; p.w = (const char*)w;
    p.x_amax = x_amax; p.w_amax = w_amax;
    p.out16 = (char*)out; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.relu = relu; p.pool = pool;
    p.ev.coef = eval_coef; p.ev.tin = eval_tin; p.ev.out_bound = out_bound; p.ev.out_tmax = out_tmax;
    return halo_dispatch(p, B, H, W, Cin, Cout, 0, "trid_conv3x3_halo_eval_p16", (hipStream_t)stream);
}

// rows per step of the ring-of-rows weight-gradient kernel, 0 when it does not cover the geometry
static int halo_wgrad_rows(int H, int W, int Cin, int Cout) {
    if (!((Cin == 32 && (Cout == 32 || Cout == 64)) || (Cin == 64 && Cout == 64)) || W <= 0 || W % 16 != 0) return 0;
    const int pg = 8 / ((Cin / 32) * (Cout / 32));
    for (int th = 4; th >= 1; th >>= 1) {
        if (H % th != 0 || (th * W / 16) % pg != 0) continue;
        const size_t xrowb = (size_t)(Cin / 32) * ((W + 2 + 7) / 8) * 1024, drowb = (size_t)(Cout / 32) * W * 128;
        if ((2 * th + 2) * xrowb + 2 * th * drowb <= 160 * 1024) return th;
    }
    return 0;
}

constexpr int HWG_SLABS = 256;

extern "C" int trid_conv3x3_wgrad_halo_rows(int H, int W, int Cin, int Cout) { return halo_wgrad_rows(H, W, Cin, Cout); }
extern "C" int trid_conv3x3_wgrad_halo_slabs(void) { return HWG_SLABS; }

template <int CIN, int COUT>
static int launch_halo_wgrad(HaloWgradParams& p, int grid, hipStream_t stream) {
    const size_t lds = std::max((size_t)(2 * p.TH + 2) * (CIN / 32) * p.xcpg * 1024 + (size_t)2 * p.TH * (COUT / 32) * p.W * 128, (size_t)8 * 32 * 33 * 4);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute((const void*)conv3x3_wgrad_halo_p16_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_err != hipSuccess) {
        set_error("trid_conv3x3_wgrad_halo_p16: cannot reserve LDS: %s", hipGetErrorString(attr_err));
        return (int)attr_err;
    }
    hipLaunchKernelGGL((conv3x3_wgrad_halo_p16_kernel<CIN, COUT>), dim3(grid), dim3(512), lds, stream, p);
    return check_launch("trid_conv3x3_wgrad_halo_p16");
}

extern "C" int trid_conv3x3_wgrad_halo_p16(const void* dy, const float* dy_amax, const void* x, const float* x_amax, float* dw, float* slabs,
                                           int B, int H, int W, int Cin, int Cout, void* stream) {
    TRID_REQUIRE(dy && x && dw && slabs && dy_amax && x_amax && B > 0 && H > 0 && W > 0, "trid_conv3x3_wgrad_halo_p16: bad arguments");
    TRID_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dw) && aligned16(slabs), "trid_conv3x3_wgrad_halo_p16: operands must be 16-byte aligned");
    const int th = halo_wgrad_rows(H, W, Cin, Cout);
    TRID_REQUIRE(th > 0, "trid_conv3x3_wgrad_halo_p16: geometry H=%d W=%d Cin=%d Cout=%d is not covered (trid_conv3x3_wgrad_halo_rows() == 0)", H, W, Cin, Cout);
    TRID_REQUIRE((long long)B * H * W * std::max(Cin, Cout) * 4 < (1ll << 31), "trid_conv3x3_wgrad_halo_p16: tensors must stay below 2 GB (31-bit buffer offsets)");
    HaloWgradParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const char*)x; p.dy = (const char*)dy; p.slabs = slabs;
    p.x_amax = x_amax; p.dy_amax = dy_amax;
    p.B = B; p.H = H; p.W = W; p.TH = th;
    p.xcpg = (W + 2 + 7) / 8;
    p.fd_xrow = make_fastdiv((uint32_t)((Cin / 32) * p.xcpg));
    p.fd_xcpg = make_fastdiv((uint32_t)p.xcpg);
    p.fd_drow = make_fastdiv((uint32_t)((Cout / 32) * (W / 8)));
    p.fd_dcpg = make_fastdiv((uint32_t)(W / 8));
    p.fdW = make_fastdiv((uint32_t)W);
    const int rg = H / th;
    int cpi = rg;  // chunks per image: the smallest divisor of the step count that gives the chip >= 256 chunks
    for (int d = 1; d <= rg; ++d)
        if (rg % d == 0 && (long long)B * d >= HWG_SLABS) {
            cpi = d;
            break;
        }
    p.chunks_per_image = cpi;
    p.rg_per_chunk = rg / cpi;
    p.nchunks = B * cpi;
    const int grid = std::min(p.nchunks, HWG_SLABS);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (Cin == 32 && Cout == 32) rc = launch_halo_wgrad<32, 32>(p, grid, s);
    else if (Cin == 32) rc = launch_halo_wgrad<32, 64>(p, grid, s);
    else rc = launch_halo_wgrad<64, 64>(p, grid, s);
    if (rc) return rc;
    return trid_slab_reduce_f32(slabs, dw, (long long)Cout * 9 * Cin, grid, (long long)Cout * 9 * Cin, 0, stream);
}

extern "C" int trid_stem_conv1_wgrad_slabs(void) { return C1W_SLABS; }

extern "C" int trid_stem_conv1_wgrad_f32(const float* img, const float* dy, float* dw, float* slabs, int B, int Hi, int Wi, void* stream) {
    TRID_REQUIRE(img && dy && dw && slabs && B > 0 && Hi > 0 && Wi > 0, "trid_stem_conv1_wgrad_f32: bad arguments");
    TRID_REQUIRE(aligned16(dw) && aligned16(slabs), "trid_stem_conv1_wgrad_f32: dw / slabs must be 16-byte aligned");
    TRID_REQUIRE((long long)B * 3 * Hi * Wi * 4 < (1ll << 31), "trid_stem_conv1_wgrad_f32: the image batch must stay below 2 GB");
    Conv1WgradParams p;
    memset(&p, 0, sizeof(p));
    p.img = img; p.dy = dy; p.slabs = slabs;
    p.B = B; p.Hi = Hi; p.Wi = Wi;
    p.Ho = (Hi + 1) / 2; p.Wo = (Wi + 1) / 2;
    p.M = (long long)B * p.Ho * p.Wo;
    TRID_REQUIRE(p.M * 128 < (1ll << 31), "trid_stem_conv1_wgrad_f32: dy must stay below 2 GB (31-bit buffer offsets)");
    const long long waves = (long long)C1W_SLABS * 4;
    p.per_wave = (int)(((p.M + waves - 1) / waves + 15) / 16 * 16);
    p.fdWo = make_fastdiv((uint32_t)p.Wo);
    p.fdHo = make_fastdiv((uint32_t)p.Ho);
    hipLaunchKernelGGL(stem_conv1_wgrad_kernel, dim3(C1W_SLABS), dim3(256), 0, (hipStream_t)stream, p);
    int rc = check_launch("trid_stem_conv1_wgrad_f32");
    if (rc) return rc;
    return trid_slab_reduce_f32(slabs, dw, 32 * 27, C1W_SLABS, 32 * 27, 0, stream);
}

static int conv1_dispatch(Conv1Params& p, int B, int Hi, int Wi, const char* who, hipStream_t stream) {
    TRID_REQUIRE((long long)B * 3 * Hi * Wi * 4 < (1ll << 31), "%s: the image batch must stay below 2 GB", who);
    p.B = B; p.Hi = Hi; p.Wi = Wi;
    p.Ho = (Hi + 1) / 2; p.Wo = (Wi + 1) / 2;
    p.M = (long long)B * p.Ho * p.Wo;
    TRID_REQUIRE(p.M * 128 < (1ll << 31), "%s: the output must stay below 2 GB (31-bit buffer offsets)", who);
    p.nslabs = (int)((p.M + 127) / 128);
    p.fdWo = make_fastdiv((uint32_t)p.Wo);
    p.fdHo = make_fastdiv((uint32_t)p.Ho);
    const int grid = std::min(p.nslabs, 256 * 8);
    if (p.out16 != nullptr) hipLaunchKernelGGL(stem_conv1_kernel<true>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(stem_conv1_kernel<false>, dim3(grid), dim3(256), 0, stream, p);
    return check_launch(who);
}

extern "C" int trid_stem_conv1_f32(const float* img, const float* w, float* y, float* stats, int B, int Hi, int Wi, void* stream) {
    TRID_REQUIRE(img && w && y && B > 0 && Hi > 0 && Wi > 0, "trid_stem_conv1_f32: bad arguments");
    Conv1Params p;
    memset(&p, 0, sizeof(p));
    p.img = img; p.w = w; p.y = y; p.stats = stats;
    return conv1_dispatch(p, B, Hi, Wi, "trid_stem_conv1_f32", (hipStream_t)stream);
}

extern "C" int trid_stem_conv1_eval_p16(const float* img, const float* w, const float* bn_scale, const float* bn_shift, void* out,
                                        const float* eval_coef, const float* eval_tin, float* out_bound, float* out_tmax, int B, int Hi, int Wi,
                                        int relu, void* stream) {
    TRID_REQUIRE(img && w && bn_scale && bn_shift && out && eval_coef && eval_tin && out_bound && B > 0 && Hi > 0 && Wi > 0 && aligned16(out),
                 "trid_stem_conv1_eval_p16: bad arguments");
    Conv1Params p;
    memset(&p, 0, sizeof(p));
    p.img = img; p.w = w;
    p.out16 = (char*)out; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.relu = relu;
    p.ev.coef = eval_coef; p.ev.tin = eval_tin; p.ev.out_bound = out_bound; p.ev.out_tmax = out_tmax;
    return conv1_dispatch(p, B, Hi, Wi, "trid_stem_conv1_eval_p16", (hipStream_t)stream);
}
