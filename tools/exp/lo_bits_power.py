#!/usr/bin/env python3
"""Does the sustained rate of the P16 GEMM depend on how many mantissa bits the LOW planes carry?  (The kernel is power-limited on
random data: zero-filled operands run 20-30 % faster at the same instruction stream.)  Same launches with the low planes of both
operands truncated to k mantissa bits (k = 10: as packed) - fewer toggling bits in two of the three products."""
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
def trunc_lo(p, keep):
    """zero the low (10 - keep) mantissa bits of every low-plane fp16 of a packed tensor (keep < 0: the whole low plane)"""
    raw = p.data.view(torch.int16).view(-1, 2, 32)  # [row * group][plane][32 channels]
    if keep < 0:
        raw[:, 1, :] = 0
    elif keep < 10:
        raw[:, 1, :] &= ~((1 << (10 - keep)) - 1)
    return p
for name, H, W, Ci, Co in [("l3.0 3x3 256 @48x16", 48, 16, 256, 256), ("l4.0 3x3 512 @24x8", 24, 8, 512, 512)]:
    M = B * H * W
    out = []
    for keep in (10, 6, 3, 0, -1):
        x, w = torch.randn(B, H, W, Ci, device=dev).relu_(), torch.randn(Co, 9 * Ci, device=dev) * 0.05
        xp, wp = trunc_lo(ops.p16_pack(x), keep), trunc_lo(ops.p16_pack(w), keep)
        y = torch.empty(B, H, W, Co, device=dev); st = ops.stats_buffer(M, Co, x)
        ms = t(lambda: ops.gemm_p16(xp, wp, y, M, Co, 9 * Ci, Co, conv=(H, W, Ci), stats=st))
        out.append("lo %2d bits %.3f ms %3.0f TF" % (keep, ms, 2.0 * M * Ci * Co * 9 / ms / 1e9))
    print(name, " | ".join(out), flush=True)
