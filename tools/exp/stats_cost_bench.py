"""GPU box: what the BatchNorm-partials epilogue costs the tile kernel: forward shapes of layer3 / layer4 with and without stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
dev = torch.device("cuda")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = [0.0, 0.0]
for (M, N, K, cnt, conv) in ((24576, 256, 1024, 12, None), (24576, 512, 2048, 6, None), (24576, 2048, 512, 6, None), (24576, 2048, 1024, 2, None), (24576, 512, 1024, 2, None),
                             (98304, 128, 512, 8, None), (98304, 256, 512, 2, None), (24576, 256, 2304, 10, (24, 8, 256)), (24576, 512, 4608, 6, (24, 8, 512)),
                             (98304, 128, 1152, 6, (48, 16, 128)), (393216, 128, 1152, 2, (96, 32, 128)), (98304, 256, 2304, 2, (48, 16, 256))):
    if conv is None:
        x = torch.randn(M, K, device=dev).relu_()
    else:
        x = torch.randn(128, conv[0], conv[1], conv[2], device=dev).relu_()
    w = torch.randn(N, K, device=dev) * 0.05
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    del x
    y = torch.empty(M, N, device=dev)
    st = torch.empty((M + 127) // 128, N, 4, device=dev)
    a = t(lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, conv=conv, stats=st, minmax=True))
    b = t(lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, conv=conv))
    tot[0] += a * cnt; tot[1] += b * cnt
    print("M %6d N %4d K %4d x%-2d %s  with partials %6.1f us, without %6.1f us (%+.1f %%)" % (M, N, K, cnt, "3x3" if conv else "1x1", a, b, 100 * (a - b) / b), flush=True)
print("per step (forward of both encoders): with %.2f ms, without %.2f ms" % (tot[0] / 1e3, tot[1] / 1e3))
