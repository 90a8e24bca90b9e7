"""GPU box: the 1x1 convolutions that stay on the tile kernel (K >= 512, and K = 256 with N <= 128) at B = 128, tile variants."""
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
variants = [int(v) for v in sys.argv[1:]] or [3, 8, 0, 1, 5]
print("default variant", ops.P16_VARIANT)
for (M, N, K, cnt) in ((24576, 256, 1024, 16), (24576, 512, 2048, 7), (24576, 2048, 512, 6), (24576, 2048, 1024, 2), (24576, 512, 1024, 3),
                       (98304, 128, 512, 10), (98304, 256, 512, 3), (24576, 1024, 512, 2), (24576, 1024, 2048, 1), (393216, 128, 256, 2)):
    x, w = torch.randn(M, K, device=dev).relu_(), torch.randn(N, K, device=dev) * 0.05
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    del x
    y = torch.empty(M, N, device=dev)
    out = "M %6d N %4d K %4d x%-2d" % (M, N, K, cnt)
    for v in variants:
        st = torch.empty((M + 127) // 128, N, 4, device=dev)
        us = t(lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, stats=st, minmax=True, variant=v))
        out += "  v%d %6.1f us (%3.0f TF)" % (v, us, 2.0 * M * N * K / us / 1e6)
    print(out, flush=True)
