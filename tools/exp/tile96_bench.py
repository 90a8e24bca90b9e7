"""GPU box: 96 x 128 tiles (variant 9) against 128 x 128 (variant 3) on the RN50 shapes at B = 128 whose 128-row grid leaves the
last round of resident workgroups partly empty (M = 24 576 with N = 256 / 512, M = 98 304 with N = 128), and on the neighbours
that fill it (controls).  Forward form (BatchNorm partials) and data-gradient form (plain store)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
dev = torch.device("cuda")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
variants = [int(v) for v in sys.argv[1:]] or [3, 9]
shapes = ((24576, 256, 1024, None, 16), (24576, 256, 2304, (24, 8, 256), 15), (24576, 512, 2048, None, 7), (24576, 512, 4608, (24, 8, 512), 9),
          (24576, 512, 1024, None, 3), (98304, 128, 512, None, 10), (98304, 128, 1152, (48, 16, 128), 9),
          (24576, 1024, 512, None, 2), (24576, 2048, 512, None, 6), (98304, 256, 2304, (48, 16, 256), 3), (393216, 128, 1152, (96, 32, 128), 3))
tot = {(v, f): 0.0 for v in variants for f in (0, 1)}
for (M, N, K, conv, cnt) in shapes:
    if conv is None:
        x = torch.randn(M, K, device=dev).relu_()
    else:
        H, W, C = conv
        x = torch.randn(M // (H * W), H, W, C, device=dev).relu_()
    w = torch.randn(N, K, device=dev) * 0.05
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    del x
    y = torch.empty(M, N, device=dev)
    out = "M %6d N %4d K %4d %-5s x%-2d auto=%3d |" % (M, N, K, "3x3" if conv else "1x1", cnt, ops.gemm_p16_rows(M, N))
    for v in variants:
        rows = ops.gemm_p16_rows(M, N, 1, v)
        st = torch.empty((M + rows - 1) // rows, N, 4, device=dev)
        us_f = t(lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, conv=conv, stats=st, minmax=True, variant=v))
        us_d = t(lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, conv=conv, variant=v))
        tot[(v, 1)] += us_f * cnt; tot[(v, 0)] += us_d * cnt
        out += "  v%d fwd %6.1f us (%3.0f TF) dgrad %6.1f us (%3.0f TF)" % (v, us_f, 2.0 * M * N * K / us_f / 1e6, us_d, 2.0 * M * N * K / us_d / 1e6)
    print(out, flush=True)
print("sum over the step's launches (count-weighted, us):", {"v%d %s" % (v, "fwd" if f else "dgrad"): round(tot[(v, f)]) for (v, f) in tot})
