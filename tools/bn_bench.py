#!/usr/bin/env python3
"""Achieved HBM bandwidth of the BatchNorm passes on the RN50 layer shapes at B=128 (isolated, sequential).
bytes = algorithmic streams (reads + writes) of each pass.  usage: python tools/bn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
B = 128
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
shapes = [("stem 192x64x64", 192, 64, 64), ("l1 bn1/2 96x32x64", 96, 32, 64), ("l1 bn3 96x32x256", 96, 32, 256),
          ("l2 bn1 96x32x128", 96, 32, 128), ("l2 bn3 48x16x512", 48, 16, 512), ("l3 bn2 24x8x256", 24, 8, 256),
          ("l3 bn3 24x8x1024", 24, 8, 1024), ("l4 bn2 24x8x512", 24, 8, 512), ("l4 bn3 24x8x2048", 24, 8, 2048)]
tot = {}
print("%-22s %10s %18s %18s %18s %18s" % ("shape", "MB/tensor", "apply", "apply+res", "bwd mode1", "bwd mode2+dres"))
for name, H, W, C in shapes:
    y = torch.randn(B, H, W, C, device=dev); g = torch.randn_like(y); res = torch.randn_like(y)
    st = ops.BNState(C, y)
    st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.uniform_(0.5, 1.5); st.shift.normal_()
    mb = y.numel() * 4 / 1e6
    out = ops.bn_apply(y, st, relu=True, res=res)
    cells = []
    for key, fn, streams in (("apply", lambda: ops.bn_apply(y, st, relu=True), 2),
                             ("apply+res", lambda: ops.bn_apply(y, st, relu=True, res=res), 3),
                             ("bwd1", lambda: ops.bn_bwd(g, y, st, None, 1), 5),
                             ("bwd2", lambda: ops.bn_bwd(g, y, st, None, 2, act=out, want_dres=True), 8)):
        ms = t(fn)
        cells.append("%6.3f ms %5.2f TB/s" % (ms, streams * mb / ms / 1e3))
        tot[key] = tot.get(key, 0) + ms
    print("%-22s %10.1f %18s %18s %18s %18s" % (name, mb, *cells))
print("sum", {k: round(v, 3) for k, v in tot.items()})
