#!/usr/bin/env python3
"""A few launches of the P16 GEMM on the big layer shapes (for rocprofv3 --pmc passes). usage: p16_prof.py <variant>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda"); B = 128; v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (H, W, Ci, Co) in ((24, 8, 512, 512), (48, 16, 256, 256)):
    x, w = torch.randn(B, H, W, Ci, device=dev).relu_(), torch.randn(Co, 9 * Ci, device=dev) * 0.05
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    y = torch.empty(B, H, W, Co, device=dev); st = ops.stats_buffer(B * H * W, Co, x)
    for _ in range(4):
        ops.gemm_p16(xp, wp, y, B * H * W, Co, 9 * Ci, Co, conv=(H, W, Ci), stats=st, variant=v)
M = B * 192
x, w = torch.randn(M, 512, device=dev).relu_(), torch.randn(2048, 512, device=dev) * 0.05
xp, wp = ops.p16_pack(x), ops.p16_pack(w)
y = torch.empty(M, 2048, device=dev); st = ops.stats_buffer(M, 2048, x)
for _ in range(4):
    ops.gemm_p16(xp, wp, y, M, 2048, 512, 2048, stats=st, variant=v)
torch.cuda.synchronize()
