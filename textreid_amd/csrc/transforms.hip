// Device-side image input pipeline: the reference's per-sample CPU chain
// (lib/data/transforms.py:4-43: Resize -> RandomHorizontalFlip -> Pad -> RandomCrop -> ToTensor -> Normalize ->
// RandomErasing, torchvision on PIL images in DataLoader workers) as two batched kernels over raw uint8 HWC images:
//   pass 1  horizontal resample of every source row to the target width (Pillow's antialiased BILINEAR: triangle
//           filter whose support scales with the down-sampling factor, 22-bit fixed-point weights, result rounded
//           to uint8 - libImaging/Resample.c; the weight tables come from the host, textreid_amd/transforms.py);
//   pass 2  vertical resample to the target height fused with flip, zero pad + crop, /255, (x - mean) / std and
//           the erase rectangle, written as fp32 NCHW - the layout the stem im2col reads.
// HBM-bound byte work: every source byte is read once, the uint8 intermediate once, the output written once.

#include "common.h"

namespace trid {

constexpr int RS_BITS = 22;  // Resample.c PRECISION_BITS for 8-bit channels

struct ImgBatch {
    const uint8_t* src;        // concatenated HWC uint8 images
    const long long* offset;   // [B] byte offset of image b in src
    const int* hw;             // [B][2] source height, width
    const int* xb;             // [B][W][2]  horizontal (first source column, count)
    const int* xk;             // [B][W][KX] horizontal weights
    const int* yb;             // [B][H][2]
    const int* yk;             // [B][H][KY]
    int B, H, W, KX, KY, maxh;
};

__global__ __launch_bounds__(256) void resample_h_kernel(ImgBatch a, uint8_t* __restrict__ tmp) {
    const int b = blockIdx.y;
    const int h = a.hw[2 * b], w = a.hw[2 * b + 1];
    const uint8_t* src = a.src + a.offset[b];
    uint8_t* dst = tmp + (long long)b * a.maxh * a.W * 3;
    const int total = h * a.W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / a.W, xo = i - y * a.W;
        const int lo = a.xb[((long long)b * a.W + xo) * 2], n = a.xb[((long long)b * a.W + xo) * 2 + 1];
        const int* kk = a.xk + ((long long)b * a.W + xo) * a.KX;
        int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
        const uint8_t* p = src + ((long long)y * w + lo) * 3;
        for (int k = 0; k < n; ++k) {
            const int c = kk[k];
            s0 += p[3 * k] * c; s1 += p[3 * k + 1] * c; s2 += p[3 * k + 2] * c;
        }
        uint8_t* o = dst + ((long long)y * a.W + xo) * 3;
        o[0] = (uint8_t)min(max(s0 >> RS_BITS, 0), 255);
        o[1] = (uint8_t)min(max(s1 >> RS_BITS, 0), 255);
        o[2] = (uint8_t)min(max(s2 >> RS_BITS, 0), 255);
    }
}

// params[b] = {flip, crop_top, crop_left, erase_i, erase_j, erase_h, erase_w, 0}
__global__ __launch_bounds__(256) void resample_v_finish_kernel(ImgBatch a, const uint8_t* __restrict__ tmp,
                                                                const int* __restrict__ params, int pad,
                                                                float m0, float m1, float m2, float d0, float d1, float d2,
                                                                float e0, float e1, float e2, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int* pr = params + 8 * b;
    const int flip = pr[0], ct = pr[1], cl = pr[2], ei = pr[3], ej = pr[4], eh = pr[5], ew = pr[6];
    const uint8_t* t = tmp + (long long)b * a.maxh * a.W * 3;
    float* ob = out + (long long)b * 3 * a.H * a.W;
    const int total = a.H * a.W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int yo = i / a.W, xo = i - yo * a.W;
        float v0, v1, v2;
        if (yo >= ei && yo < ei + eh && xo >= ej && xo < ej + ew) {
            v0 = e0; v1 = e1; v2 = e2;  // RandomErasing value, written after normalisation (transforms.py:24)
        } else {
            const int yy = yo + ct - pad, xr = xo + cl - pad;  // position in the resized (and flipped) image
            int u0 = 0, u1 = 0, u2 = 0;                          // T.Pad fills with 0
            if (yy >= 0 && yy < a.H && xr >= 0 && xr < a.W) {
                const int xx = flip ? a.W - 1 - xr : xr;
                const int h = a.hw[2 * b];
                if (h == a.H) {  // Pillow skips the vertical pass when the height already matches
                    const uint8_t* p = t + ((long long)yy * a.W + xx) * 3;
                    u0 = p[0]; u1 = p[1]; u2 = p[2];
                } else {
                    const int lo = a.yb[((long long)b * a.H + yy) * 2], n = a.yb[((long long)b * a.H + yy) * 2 + 1];
                    const int* kk = a.yk + ((long long)b * a.H + yy) * a.KY;
                    int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
                    const uint8_t* p = t + ((long long)lo * a.W + xx) * 3;
                    for (int k = 0; k < n; ++k) {
                        const int c = kk[k];
                        const uint8_t* q = p + (long long)k * a.W * 3;
                        s0 += q[0] * c; s1 += q[1] * c; s2 += q[2] * c;
                    }
                    u0 = min(max(s0 >> RS_BITS, 0), 255);
                    u1 = min(max(s1 >> RS_BITS, 0), 255);
                    u2 = min(max(s2 >> RS_BITS, 0), 255);
                }
            }
            v0 = ((float)u0 / 255.f - m0) / d0;  // ToTensor then Normalize, IEEE division like torch's div_
            v1 = ((float)u1 / 255.f - m1) / d1;
            v2 = ((float)u2 / 255.f - m2) / d2;
        }
        ob[i] = v0;
        ob[(long long)total + i] = v1;
        ob[2LL * total + i] = v2;
    }
}

}  // namespace trid

using namespace trid;

extern "C" long long trid_image_pipeline_ws_bytes(int B, int max_src_h, int W) {
    return (long long)B * max_src_h * W * 3;
}

extern "C" int trid_image_pipeline_u8(const uint8_t* src, const long long* offset, const int* hw, const int* xbounds,
                                      const int* xweights, const int* ybounds, const int* yweights,
                                      const int* params, int B, int H, int W, int KX, int KY, int max_src_h, int pad,
                                      const float* mean3_std3_erase3_host, uint8_t* ws, float* out, void* stream) {
    TRID_REQUIRE(src && offset && hw && xbounds && xweights && ybounds && yweights && params && ws && out &&
                     mean3_std3_erase3_host,
                 "trid_image_pipeline_u8: null pointer");
    TRID_REQUIRE(B > 0 && H > 0 && W > 0 && KX > 0 && KY > 0 && max_src_h > 0 && pad >= 0,
                 "trid_image_pipeline_u8: bad sizes");
    ImgBatch a{src, offset, hw, xbounds, xweights, ybounds, yweights, B, H, W, KX, KY, max_src_h};
    const float* c = mean3_std3_erase3_host;  // HOST array of 9 floats (configuration constants)
    TRID_REQUIRE(c[3] != 0.f && c[4] != 0.f && c[5] != 0.f, "trid_image_pipeline_u8: zero std");
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((long long)max_src_h * W, 256, 256), B), dim3(256), 0, st, a, ws);
    hipLaunchKernelGGL(resample_v_finish_kernel, dim3(grid_for((long long)H * W, 256, 256), B), dim3(256), 0, st, a, ws,
                       params, pad, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], out);
    return check_launch("trid_image_pipeline_u8");
}
