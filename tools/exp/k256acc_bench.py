"""GPU box: layer3 / layer4's conv1 data gradient shapes with accumulate (K = 256 / 512): tile kernel against the streaming kernel."""
import os, sys
import torch as T
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
dev = T.device("cuda")
def t(fn, reps=20):
    fn(); T.cuda.synchronize()
    e0, e1 = T.cuda.Event(enable_timing=True), T.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); T.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in ((24576, 1024, 256), (98304, 512, 256)):
    x, w = T.randn(M, K, device=dev), T.randn(N, K, device=dev) * 0.2
    xp, wp = ops.p16_pack(x), ops.p16_pack(w)
    c = T.randn(M, N, device=dev)
    keep = T.rand(M, N, device=dev) < 0.6
    for stream in (False, True, False, True):
        ops.USE_STREAM = stream
        print("M %d N %d K %d %s: accumulate %.1f us" % (M, N, K, "stream" if stream else "tile  ", t(lambda: ops.gemm_p16(xp, wp, c, M, N, K, N, accumulate=True))), flush=True)
