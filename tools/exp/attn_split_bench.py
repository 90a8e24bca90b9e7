"""GPU box: the attention pool's per-image score product S[b] = U[b] (32 x 2048) . tok[b]^T (2048 x 196) at B = 128 with split K."""
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
B, heads, C, T1p = 128, 32, 2048, 196
U, tok = T.randn(B, heads, C, device=dev), T.randn(B, T1p, C, device=dev)
ref = T.einsum("bhc,btc->bht", U.double(), tok.double())
for sp in (1, 2, 4, 8):
    P = ops.empty((B, heads, T1p), U)
    slab = ops.empty((sp, B, heads, T1p), U) if sp > 1 else None
    def run():
        if sp == 1:
            ops.gemm(U, tok, P, heads, T1p, C, C, C, T1p, batch=B, strideA=heads * C, strideB=T1p * C, strideC=heads * T1p)
        else:
            ops.gemm(U, tok, slab, heads, T1p, C, C, C, T1p, batch=B, strideA=heads * C, strideB=T1p * C, strideC=heads * T1p, splits=sp, strideSplit=B * heads * T1p)
            ops.call("trid_slab_reduce_f32", ops._p(slab), ops._p(P), B * heads * T1p, sp, B * heads * T1p, 0, ops.stream())
    us = t(run)
    print("splits %d: %.1f us  err %.1e" % (sp, us, float((P.double() - ref).abs().max() / ref.abs().max())), flush=True)
