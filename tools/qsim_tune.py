"""Timing of the fused queue-similarity / InfoNCE block (queue_nce.hip) against the unfused form and
over workgroup counts.  GPU box:  python tools/qsim_tune.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import losses as L

dev = torch.device("cuda")
B, C = 128, 256
g = torch.Generator().manual_seed(1)
nrm = lambda t: torch.nn.functional.normalize(t, dim=1).to(dev)
vq, tq, vk, tk = (nrm(torch.randn(B, C, generator=g)) for _ in range(4))
ids = torch.arange(B, device=dev) // 4


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for K in (8192, 65536):
    tqueue, vqueue = nrm(torch.randn(K, C, generator=g)), nrm(torch.randn(K, C, generator=g))
    idq = torch.randint(0, 11003, (1, K), generator=g).to(dev)
    nbytes = 2 * C * K * 4 + 8 * K + 4 * B * C * 4 + 2 * B * C * 4
    fn = lambda: L.queue_infonce_loss(vq, tq, vk, tk, ids, tqueue, vqueue, idq)
    L.FUSED_QUEUE_NCE = False
    ms = timeit(fn)
    print("K=%6d unfused            %.3f ms  %.0f GB/s" % (K, ms, nbytes / ms / 1e6))
    L.FUSED_QUEUE_NCE = True
    for wgs in (16, 32, 64, 128, 256, 512):
        L.QUEUE_NCE_WGS = wgs
        ms = timeit(fn)
        print("K=%6d fused wgs/mod %4d %.3f ms  %.0f GB/s  (%.1f TFLOP/s fp32-equivalent)" % (K, wgs, ms, nbytes / ms / 1e6, 8.0 * B * C * K / ms / 1e9))
    L.QUEUE_NCE_WGS = 0
    for fused_sum in (False, True):
        L.FUSED_QUEUE_SUM = fused_sum
        ms = timeit(fn, reps=200)
        print("K=%6d default workgroups, loss rows folded by %s: %.4f ms  %.0f GB/s" % (K, "the finish launch" if fused_sum else "trid_sum_f32", ms, nbytes / ms / 1e6))
