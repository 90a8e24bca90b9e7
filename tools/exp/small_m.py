#!/usr/bin/env python3
"""Tile quantisation on the small-M layers (layer3 / layer4 at B=128): 128x128 tiles (variant 3) vs 128x64 (variant 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreid_amd import ops
dev = torch.device("cuda"); B = 128
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def run(name, H, W, Ci, Co, conv3, stats=True):
    M = B * H * W
    x = torch.randn(B, H, W, Ci, device=dev).relu_()
    w = torch.randn(Co, (9 if conv3 else 1) * Ci, device=dev) * 0.05
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    y = torch.empty(B, H, W, Co, device=dev); st = ops.stats_buffer(M, Co, x) if stats else None
    out = []
    for v in (3, 8):
        ms = t(lambda: ops.gemm_p16(xp, wp, y, M, Co, (9 if conv3 else 1) * Ci, Co, conv=(H, W, Ci) if conv3 else None, stats=st, variant=v))
        out.append("v%d %.3f ms" % (v, ms))
    tiles = ((M + 127) // 128) * ((Co + 127) // 128)
    print("%-34s tiles(128x128) %4d  %s" % (name, tiles, "  ".join(out)), flush=True)
run("l3 conv1 1024->256 @24x8", 24, 8, 1024, 256, False)
run("l3 conv2 3x3 256 @24x8", 24, 8, 256, 256, True)
run("l3 conv3 256->1024 @24x8", 24, 8, 256, 1024, False)
run("l3 dgrad conv3 1024->256 @24x8", 24, 8, 1024, 256, False, stats=False)
run("l4.0 conv1 1024->512 @24x8", 24, 8, 1024, 512, False)
run("l4 conv1 2048->512 @12x4", 12, 4, 2048, 512, False)
run("l4 conv2 3x3 512 @12x4", 12, 4, 512, 512, True)
run("l4 conv3 512->2048 @12x4", 12, 4, 512, 2048, False)
run("l4 dgrad conv1 512->2048 @12x4", 12, 4, 512, 2048, False, stats=False)
run("l2 conv2 3x3 128 @48x16", 48, 16, 128, 128, True)
run("l2 conv1 512->128 @48x16", 48, 16, 512, 128, False)
