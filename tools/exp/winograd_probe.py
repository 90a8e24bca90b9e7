"""GPU box: go / no-go numbers for a Winograd F(2x2, 3x3) form of layer4.conv2 (M = 24 576 pixels, 512 -> 512 channels, 24 x 8 maps,
128 images).  F(2x2, 3x3) turns the convolution into 16 independent [tiles x Cin] x [Cin x Cout] products over the
6 144 2x2 output tiles (2.25x fewer multiply-adds) plus an input transform (4x4 patches, 4x the activation bytes) and an output
transform.  Measured here: the direct kernel, and the 16-fold batched GEMM that is the LOWER bound of any Winograd form on this
library's tile kernel; the transform passes are priced at the bytes they move."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
from textreid_amd.lib import GemmDesc, call
dev = torch.device("cuda")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, H, W, C = 128, 24, 8, 512
x = torch.randn(B, H, W, C, device=dev).relu_()
w = torch.randn(C, 9 * C, device=dev) * 0.02
xp, wp = ops.p16_pack(x), ops.p16_pack(w)
y = torch.empty(B * H * W, C, device=dev)
us_direct = t(lambda: ops.gemm_p16(xp, wp, y, B * H * W, C, 9 * C, C, conv=(H, W, C)))
T = B * (H // 2) * (W // 2)  # 2x2 output tiles
V = ops.p16_pack(torch.randn(16 * T, C, device=dev))   # transformed input patches, one [T, Cin] matrix per transform position
U = ops.p16_pack(torch.randn(16 * C, C, device=dev) * 0.02)  # transformed filters [16][Cout][Cin]
Mo = torch.empty(16 * T, C, device=dev)
d = GemmDesc()
d.A, d.B, d.C = V.data.data_ptr(), U.data.data_ptr(), Mo.data_ptr()
d.M, d.N, d.K = T, C, C
d.lda, d.ldb, d.ldc = C, C, C
d.batch, d.splits = 16, 1
d.strideA, d.strideB, d.strideC = T * C, C * C, T * C
d.a_mode, d.b_mode, d.alpha, d.precision = 0, 0, 1.0, 16
d.a_amax, d.b_amax = V.amax.data_ptr(), U.amax.data_ptr()
us_batched = t(lambda: call("trid_gemm_p16", ctypes.addressof(d), -1, ops.stream()))
fl = 2.0 * B * H * W * C * 9 * C
mb_in = (B * H * W * C * 4 + 16 * T * C * 4) / 1e6        # read the P16 activation, write 16 transformed P16 matrices
mb_out = (16 * T * C * 4 + B * H * W * C * 4) / 1e6       # read 16 fp32 product matrices, write the fp32 output
print("direct 3x3 implicit GEMM        : %6.1f us  (%3.0f TFLOP/s fp32-equivalent)" % (us_direct, fl / us_direct / 1e6))
print("16 x [%d x %d] x [%d x %d] batched: %6.1f us  (the multiply-adds of F(2x2,3x3): %.2fx fewer)" % (T, C, C, C, us_batched, fl / (16 * 2.0 * T * C * C)))
print("transform passes, priced at 5 TB/s: input %5.1f MB = %5.1f us, output %5.1f MB = %5.1f us" % (mb_in, mb_in / 5.0, mb_out, mb_out / 5.0))
print("Winograd with separate transform passes >= %6.1f us;  with both transforms fused into the GEMM kernel >= %6.1f us;  go / no-go bar 200 us" % (
    us_batched + mb_in / 5.0 + mb_out / 5.0, us_batched))
