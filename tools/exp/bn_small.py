#!/usr/bin/env python3
"""BatchNorm passes of the P16 / bf16 data flow on the mid / small layer shapes: time vs the HBM time of their bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreid_amd import ops
dev = torch.device("cuda"); B = 128
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("%-20s %8s | %s" % ("shape", "MB fp32", "us (ideal us at 5 TB/s): fwd apply P16 | bwd P16 (reduce+final+apply) | bwd bf16, g fp32 | bwd bf16, g bf16"))
for name, H, W, C in [("l2 bn1 96x32x128", 96, 32, 128), ("l2 bn2 48x16x128", 48, 16, 128), ("l2 bn3 48x16x512", 48, 16, 512), ("l3 bn1 48x16x256", 48, 16, 256),
                      ("l3 bn2 24x8x256", 24, 8, 256), ("l3 bn3 24x8x1024", 24, 8, 1024), ("l4 bn2 12x4x512", 12, 4, 512), ("l4 bn3 12x4x2048", 12, 4, 2048)]:
    y = torch.randn(B, H, W, C, device=dev); g = torch.randn_like(y)
    st = ops.BNState(C, y)
    st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.uniform_(0.5, 1.5); st.shift.normal_()
    mb = y.numel() * 4 / 1e6
    bound = ops.amax(y)
    g16 = g.to(torch.bfloat16)
    cells = []
    for fn, nbytes in ((lambda: ops.bn_apply_p16(y, st, bound, relu=True), 2 * mb),
                       (lambda: ops.bn_bwd_p16(g, y, st, 1, fmt=1), 5 * mb),
                       (lambda: ops.bn_bwd_p16(g, y, st, 1, fmt=2), 4.5 * mb),
                       (lambda: ops.bn_bwd_p16(g16, y, st, 1, fmt=2), 3.5 * mb)):
        us = t(fn)
        cells.append("%6.1f (%5.1f)" % (us, nbytes / 5.0))
    print("%-20s %8.1f | %s" % (name, mb, " | ".join(cells)), flush=True)
