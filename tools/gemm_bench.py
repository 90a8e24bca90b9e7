#!/usr/bin/env python3
"""Micro-benchmark of the GEMM / implicit-conv kernel on the RN50 layer shapes (B=128).
usage: python tools/gemm_bench.py [filter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops

dev = torch.device("cuda")
flt = sys.argv[1] if len(sys.argv) > 1 else ""

def timeit(fn, flops, name, bytes_=0, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-44s %8.3f ms  %6.1f TF/s  %6.2f TB/s" % (name, ms, flops / ms / 1e9, bytes_ / ms / 1e9), flush=True)

def rnd(*s): return torch.randn(*s, device=dev)

cases = []
def add(name, fn, flops, bytes_=0):
    if flt in name: cases.append((name, fn, flops, bytes_))

B = 128
# plain GEMM calibration
for n in (2048, 4096):
    a, w = rnd(n, n), rnd(n, n); out = torch.empty(n, n, device=dev)
    add("nt %d^3" % n, lambda a=a, w=w, out=out: ops.linear(a, w, out=out), 2.0 * n ** 3, 3 * 4 * n * n)
# 3x3 convs (fwd): (H, W, C)
for (H, W, C) in [(192, 64, 32), (96, 32, 64), (96, 32, 128), (48, 16, 128), (48, 16, 256), (24, 8, 256), (24, 8, 512)]:
    x, w = rnd(B, H, W, C), rnd(C, 9 * C)
    M = B * H * W
    add("conv3x3 fwd %dx%d C%d" % (H, W, C), lambda x=x, w=w: ops.conv3x3(x, w, stats=True), 2.0 * M * C * 9 * C, 4 * 2 * M * C)
    dy = rnd(B, H, W, C)
    add("conv3x3 wgrad %dx%d C%d" % (H, W, C), lambda x=x, dy=dy: ops.conv3x3_wgrad(dy, x), 2.0 * M * C * 9 * C, 4 * 2 * M * C)
# 1x1 convs: (pixels per image, Cin, Cout)
for (P, Ci, Co) in [(3072, 64, 256), (3072, 256, 64), (768, 512, 128), (768, 128, 512), (192, 1024, 256), (192, 256, 1024), (192, 2048, 512), (192, 512, 2048), (192, 1024, 2048)]:
    M = B * P
    x, w = rnd(M, Ci), rnd(Co, Ci)
    add("conv1x1 fwd M%d %d->%d" % (M, Ci, Co), lambda x=x, w=w: ops.conv1x1(x, w, stats=True), 2.0 * M * Ci * Co, 4 * M * (Ci + Co))
    dy = rnd(M, Co); dx = torch.empty(M, Ci, device=dev)
    add("conv1x1 dgrad M%d %d<-%d" % (M, Ci, Co), lambda dy=dy, w=w, dx=dx: ops.matmul_nn(dy, w, out=dx), 2.0 * M * Ci * Co, 4 * M * (Ci + Co))
    add("conv1x1 dgrad+acc M%d %d<-%d" % (M, Ci, Co), lambda dy=dy, w=w, dx=dx: ops.matmul_nn(dy, w, out=dx, accumulate=True), 2.0 * M * Ci * Co, 4 * M * (2 * Ci + Co))
    add("conv1x1 wgrad M%d %d->%d" % (M, Ci, Co), lambda dy=dy, x=x: ops.conv1x1_wgrad(dy, x), 2.0 * M * Ci * Co, 4 * M * (Ci + Co))
for name, fn, fl, by in cases:
    timeit(fn, fl, name, by)
