#!/usr/bin/env python3
"""Is the P16 3x3 GEMM power-limited?  Same launches on random, constant and zero-filled operands (MFMA timing does not
depend on the data, the power drawn - hence the sustained clock - does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreid_amd import ops
dev = torch.device("cuda"); B = 128
def t(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, H, W, Ci, Co in [("l2.0 3x3 128 @96x32", 96, 32, 128, 128), ("l3.0 3x3 256 @48x16", 48, 16, 256, 256), ("l4.0 3x3 512 @24x8", 24, 8, 512, 512)]:
    M = B * H * W
    out = []
    for kind in ("random", "ones", "zeros"):
        if kind == "random":
            x, w = torch.randn(B, H, W, Ci, device=dev).relu_(), torch.randn(Co, 9 * Ci, device=dev) * 0.05
        elif kind == "ones":
            x, w = torch.ones(B, H, W, Ci, device=dev), torch.ones(Co, 9 * Ci, device=dev)
        else:
            x, w = torch.zeros(B, H, W, Ci, device=dev), torch.zeros(Co, 9 * Ci, device=dev)
        one = torch.ones(1, device=dev)
        xp, wp = ops.p16_pack(x, one), ops.p16_pack(w, one)
        y = torch.empty(B, H, W, Co, device=dev); st = ops.stats_buffer(M, Co, x)
        for v in (3, 6):
            ms = t(lambda: ops.gemm_p16(xp, wp, y, M, Co, 9 * Ci, Co, conv=(H, W, Ci), stats=st, variant=v))
            out.append("%s v%d %.3f ms %3.0f TF" % (kind, v, ms, 2.0 * M * Ci * Co * 9 / ms / 1e9))
    print(name, " | ".join(out), flush=True)
