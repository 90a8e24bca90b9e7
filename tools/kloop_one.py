#!/usr/bin/env python3
"""One shape of tools/kloop_bench.py, one variant, a few launches - the target of rocprofv3 --pmc passes.
usage: python tools/kloop_one.py <H> <W> <C> <variant> [zero] [reps]     (3x3 conv at B=128; N = C, K = 9C)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
H, W, C, v = (int(a) for a in sys.argv[1:5])
zero = len(sys.argv) > 5 and sys.argv[5] == "1"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
B = 128
M, N, K = B * H * W, C, 9 * C
x = torch.zeros(B, H, W, C, device=dev) if zero else torch.randn(B, H, W, C, device=dev).relu_()
w = torch.zeros(N, K, device=dev) if zero else torch.randn(N, K, device=dev) * 0.05
ax, aw = ops.amax(x), ops.amax(w)
if zero:
    ax.fill_(1.0); aw.fill_(1.0)
xp, wp = ops.p16_pack(x, ax), ops.p16_pack(w, aw)
y = torch.empty(M, N, device=dev)
st = torch.zeros((M + 127) // 128, N, 4, device=dev)
for _ in range(reps):
    ops.gemm_p16(xp, wp, y, M, N, K, N, conv=(H, W, C), stats=st, variant=v, minmax=True)
torch.cuda.synchronize()
