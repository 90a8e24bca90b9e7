"""GPU box: bn_finalize_minmax at the step's (parts, rows per part, M, C) shapes; us per call (events over 20 calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops  # noqa: E402

dev = torch.device("cuda")
for (nparts, rpp, M, C) in ((3072, 128, 393216, 64), (3072, 128, 393216, 256), (3072, 128, 393216, 128), (192, 128, 24576, 2048),
                            (384, 64, 24576, 1024), (192, 128, 24576, 256), (192, 128, 24576, 512), (768, 128, 98304, 512),
                            (768, 128, 98304, 128), (768, 128, 98304, 256), (12288, 128, 1572864, 32), (12288, 128, 1572864, 64), (6144, 256, 1572864, 32)):
    st = torch.randn(nparts, C, 4, device=dev)
    st[..., 1].abs_()
    st[..., 2] = -1.0
    st[..., 3] = 1.0
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    bound = torch.zeros(1, device=dev)
    run = lambda: ops.bn_finalize_minmax(st, M, g, b, None, None, True, bound, rows_per_part=rpp)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("parts %6d x %3d rows, C %5d: %6.1f us" % (nparts, rpp, C, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
