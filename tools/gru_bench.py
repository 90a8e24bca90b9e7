#!/usr/bin/env python3
"""Text encoder alone (B=128, H=E=512, 64 time steps): fused one-launch-per-step kernels (gru_step.hip) against the
GEMM + cell-kernel form.  GPU box:  python tools/gru_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd.backbones import gru as G
from textreid_amd.caption import CaptionBatch

dev = torch.device("cuda")
B, H, L, vocab = 128, 512, 64, 1000
g = torch.Generator().manual_seed(0)
m = G.GRU(H, H, H, 1, 0.0, True, "clip_vit", "./", vocab_dict=torch.randn(vocab, H, generator=g) * 0.5).to(dev)
lengths = torch.randint(20, L + 1, (B,), generator=g); lengths[0] = L
cb = CaptionBatch(torch.randint(0, vocab, (B, L), generator=g).to(dev), lengths.to(dev), max_len=L)
gout = torch.randn(B, 2 * H, generator=g).to(dev)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fwd():
    with torch.no_grad():
        m(cb)


def fwd_bwd():
    m.zero_grad()
    (m(cb) * gout).sum().backward()


for fused in (False, True):
    G.FUSED_GRU_STEP = fused
    print("fused=%d  forward %.3f ms   forward+backward %.3f ms" % (fused, timeit(fwd), timeit(fwd_bwd)))
