"""GPU box: the residual blocks' short-K 1x1 convolutions at B = 128 (RN50 layer shapes), the tile kernel gemm_p16_kernel<A_KC>
against the streaming kernel of csrc/gemm_stream.hip: microseconds per launch and HBM rate of the algorithmic bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreid_amd import ops  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# (name, M, K, N, accumulate)
SHAPES = [("l1 conv3 / downsample 64->256", 393216, 64, 256, False), ("l1.0 conv1 64->64", 393216, 64, 64, False),
          ("l2 conv3 128->512", 98304, 128, 512, False), ("l2.0 conv1 256->128", 393216, 256, 128, False),
          ("l2.0 downsample 256->512", 98304, 256, 512, False), ("l3 conv3 256->1024", 24576, 256, 1024, False),
          ("l1.0 dgrad conv1 64->64 (+=)", 393216, 64, 64, True), ("l1.x dgrad conv1 64->256 (+=)", 393216, 64, 256, True),
          ("l2.0 dgrad conv1 128->256 (+=)", 393216, 128, 256, True), ("l2.x dgrad conv1 128->512 (+=)", 98304, 128, 512, True)]
tot = [0.0, 0.0]
for name, M, K, N, acc in SHAPES:
    x = torch.relu(torch.randn(M, K, generator=g)).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).to(dev)
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    out = torch.zeros(M, N, device=dev)
    nb = M * (K + N * (2 if acc else 1)) * 4
    res = []
    for stream in (False, True):
        ops.USE_STREAM = stream
        if acc:
            us = t(lambda: ops.gemm_p16(xp, wp, out, M, N, K, N, accumulate=True))
        else:
            us = t(lambda: ops.conv_p16(xp, wp))
        res.append(us)
    tot[0] += res[0]
    tot[1] += res[1]
    print("%-34s M=%6d  tile %7.1f us (%5.2f TB/s)   streaming %7.1f us (%5.2f TB/s)   %4.0f MB" % (
        name, M, res[0], nb / res[0] / 1e6, res[1], nb / res[1] / 1e6, nb / 1e6), flush=True)
    del x, xp, out
print("sum: tile %.1f us, streaming %.1f us" % tuple(tot))
