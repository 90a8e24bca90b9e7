#!/usr/bin/env python3
"""Ablation timing of the P16 3x3 / 1x1 forward GEMM: the library under TRID_LIB_PATH is built with the DMA, the LDS
reads or the MFMAs removed (results are garbage; only the time matters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
B = 128
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = []
for name, H, W, Ci, Co in [("l1", 96, 32, 64, 64), ("l2", 48, 16, 128, 128), ("l3", 24, 8, 256, 256), ("l4", 12, 4, 512, 512)]:
    x, w = torch.randn(B, H, W, Ci, device=dev).relu_(), torch.randn(Co, 9 * Ci, device=dev) * 0.05
    xp, wp = ops.p16_pack(x, ops.amax(x)), ops.p16_pack(w, ops.amax(w))
    M = B * H * W
    y = torch.empty(B, H, W, Co, device=dev); st = ops.stats_buffer(M, Co, x)
    for stats in (st, None):
        ms = t(lambda: ops.gemm_p16(xp, wp, y, M, Co, 9 * Ci, Co, conv=(H, W, Ci), stats=stats, variant=3))
        out.append("%s%s %.3f/%3.0fTF" % (name, "s" if stats is not None else "", ms, 2.0 * M * Ci * Co * 9 / ms / 1e9))
print(os.environ.get("TRID_LIB_PATH", "product"), " ".join(out), flush=True)
