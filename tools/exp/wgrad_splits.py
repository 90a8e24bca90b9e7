#!/usr/bin/env python3
"""Split-count sweep of the P16 weight-gradient GEMM (with / without the XCD-owns-split mapping: TRID_WGRAD_XCD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreid_amd import ops
dev = torch.device("cuda"); B = 128
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def run(name, H, W, Ci, Co, conv, cands):
    M = B * H * W
    x = torch.randn(B, H, W, Ci, device=dev).relu_() if conv else torch.randn(M, Ci, device=dev).relu_()
    dy = torch.randn(B, H, W, Co, device=dev) if conv else torch.randn(M, Co, device=dev)
    xp, dyp = ops.p16_pack(x, ops.amax(x)), ops.p16_pack(dy, ops.amax(dy))
    out = []
    for s in cands:
        ops._WGRAD_SPLITS_OVERRIDE = s
        ms = t(lambda: ops.wgrad_p16(dyp, xp, conv=(H, W, Ci) if conv else None))
        out.append("%s:%.3f" % (s, ms))
    ops._WGRAD_SPLITS_OVERRIDE = None
    print("%-26s %s" % (name, " ".join(out)), flush=True)
print("XCD", os.environ.get("TRID_WGRAD_XCD", "0"))
run("l1 3x3 64 @96x32", 96, 32, 64, 64, True, [None, 96, 128, 160, 200, 208, 256])
run("l2.0 3x3 128 @96x32", 96, 32, 128, 128, True, [None, 56, 64, 88, 112, 120, 168])
run("l2 3x3 128 @48x16", 48, 16, 128, 128, True, [None, 56, 64, 88, 112, 120, 168])
run("l3.0 3x3 256 @48x16", 48, 16, 256, 256, True, [None, 16, 24, 32, 40, 56])
run("l3 3x3 256 @24x8", 24, 8, 256, 256, True, [None, 16, 24, 32, 40, 48])
run("l4 3x3 512 @24x8", 24, 8, 512, 512, True, [None, 4, 7, 8, 16])
run("l4 3x3 512 @12x4", 12, 4, 512, 512, True, [None, 4, 7, 8, 12])
run("l1 conv3 64->256", 96, 32, 64, 256, False, [None, 128, 256, 384, 512])
run("l2 conv3 128->512", 48, 16, 128, 512, False, [None, 64, 128, 192])
run("l3 conv3 256->1024", 24, 8, 256, 1024, False, [None, 16, 32, 48])
run("l4 conv3 512->2048", 12, 4, 512, 2048, False, [None, 8, 12, 16])
run("l4 conv1 2048->512", 12, 4, 2048, 512, False, [None, 8, 12, 16])
