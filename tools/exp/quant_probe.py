"""GPU box: is the M = 24 576 x N = 256 tile GEMM slow because of its grid (384 tiles on 512 resident slots)?  The same kernel at
256 / 384 / 512 / 768 / 1024 tiles: time per tile."""
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
for N, K in ((256, 1024), (256, 2304), (512, 2048)):
    for M in (16384, 24576, 32768, 49152, 65536):
        x, w = torch.randn(M, K, device=dev).relu_(), torch.randn(N, K, device=dev) * 0.05
        xp, wp = ops.p16_pack(x), ops.p16_pack(w)
        y = torch.empty(M, N, device=dev)
        us = t(lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, variant=3))
        tiles = (M // 128) * (N // 128)
        print("N %4d K %4d M %6d: %4d tiles  %6.1f us  %5.3f us/tile  %4.0f TF" % (N, K, M, tiles, us, us / tiles, 2.0 * M * N * K / us / 1e6), flush=True)
