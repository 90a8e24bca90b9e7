#!/usr/bin/env python3
"""Cycle-level probe of the ping-pong GEMM (library built with -DTRID_PP_TRACE, see gemm_pp.hip):
per K-tile durations of the staging role (split+store, load issue), barrier waits and the compute role
for the two wave groups of workgroup 0.  usage: TRID_LIB_PATH=.../lib_exp_trace.so TRID_PP=1 python tools/pp_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
M, N, K = 24576, 2048, 1024
conv = len(sys.argv) > 1 and sys.argv[1] == "conv"
if conv:
    x = torch.randn(128, 24, 8, 512, device=dev); w = torch.randn(512, 9 * 512, device=dev)
else:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
trace = torch.zeros(2 * 64 * 8, dtype=torch.int64, device=dev)
fake_bias = trace.view(torch.float32)
for _ in range(3):
    trace.zero_()
    if conv:
        y = ops.empty((128, 24, 8, 512), x)
        ops.gemm(x, w, y, 128 * 24 * 8, 512, 9 * 512, 512, 9 * 512, 512, a_mode=ops.A_CONV, conv=(24, 8, 512), bias=fake_bias)
    else:
        y = ops.empty((M, N), x)
        ops.gemm(x, w, y, M, N, K, K, K, N, bias=fake_bias)
    torch.cuda.synchronize()
t = trace.cpu().view(2, 64, 8)
print("events: 0 stage start, 1 after split+LDS store, 2 after load issue, 3 after the barrier that ends the staging half-step, 4 compute start, 5 compute end")
for g in range(2):
    print("group", g)
    for kt in range(2, 14):
        e = t[g, kt]
        print("  kt %2d: vmcnt-wait %5d  split+store %5d  load-issue %5d  wait-barrier(after stage) %5d | compute %5d | period %5d" % (
            kt, e[6] - e[0], e[1] - e[6], e[2] - e[1], e[3] - e[2], e[5] - e[4], t[g, kt + 1, 4] - e[4]))
