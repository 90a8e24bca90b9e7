"""GPU box: the attention pool's products at B = 128 (RN50: 32 heads, 193 -> 196 tokens, 2048 channels) on the tiled kernels and on
the skinny kernel, microseconds each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops

dev = torch.device("cuda")
B, heads, C, T1p = 128, 32, 2048, 196
hd = C // heads
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)


def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tok, U, P, Z, q, o = r(B, T1p, C), r(B, heads, C), r(B, heads, T1p), r(B, heads, C), r(B, C), r(B, C)
Wq, Wv, Wc, bq = r(C, C), r(C, C), r(1024, C), r(C)
Pout, Zout, oout = torch.empty(B, heads, T1p, device=dev), torch.empty(B, heads, C, device=dev), torch.empty(B, C, device=dev)
cases = [
    ("q_proj  linear [128x2048x2048]", lambda: ops.linear(tok[:, 0], Wq, bq)),
    ("c_proj  linear [128x1024x2048]", lambda: ops.linear(o, Wc)),
    ("embed   linear [128x256x1024]", lambda: ops.linear(o[:, :1024].contiguous(), Wc[:256, :1024].contiguous())),
    ("S = U tok^T   batch 128 [32x196x2048]", lambda: ops.gemm(U, tok, Pout, heads, T1p, C, C, C, T1p, batch=B, strideA=heads * C, strideB=T1p * C, strideC=heads * T1p)),
    ("Z = P tok     batch 128 [32x2048x196]", lambda: ops.gemm(P, tok, Zout, heads, C, T1p, T1p, C, C, b_mode=ops.B_NC, batch=B, strideA=heads * T1p, strideB=T1p * C, strideC=heads * C)),
    ("o = Z Wv^T    batch 32  [128x64x2048]", lambda: ops.gemm(Z, Wv, oout, B, hd, C, heads * C, C, C, batch=heads, strideA=C, strideB=hd * C, strideC=hd)),
    ("do = gout Wc  matmul_nn [128x2048x1024]", lambda: ops.matmul_nn(o[:, :1024].contiguous(), Wc)),
]
for name, fn in cases:
    res = []
    for sk in (False, True):
        ops.USE_SKINNY = sk
        res.append(t(fn))
    print("%-42s tiled %7.1f us   skinny %7.1f us" % (name, res[0], res[1]), flush=True)
