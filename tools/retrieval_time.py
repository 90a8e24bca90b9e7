"""GPU box: configs[4] match (Q = 1e4 queries x G = 1e6 gallery rows, top-10) - pre-split operands + streaming filter kernel against
the on-the-fly split GEMM; agreement of the results.  usage: python tools/retrieval_time.py [G] [Q]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import textreid_amd.evaluation as E
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
gen = torch.Generator(device="cpu").manual_seed(7)
q = torch.nn.functional.normalize(torch.randn(Q, 256, generator=gen), dim=1).cuda()
g = torch.nn.functional.normalize(torch.randn(G, 256, generator=gen), dim=1).cuda()
res = {}
ONLY = os.environ.get("TRID_RETR_ONLY_P16", "0") == "1"  # (profiling runs: 1 warm-up + 3 timed calls of the product path)
for name, flag in ((("pre-split + streaming filter", True),) if ONLY else (("on-the-fly split", False), ("pre-split + streaming filter", True)) * 2):
    E.USE_SIM_P16 = flag
    E.similarity_topk(q, g, 10, normalize=False); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        res[name] = E.similarity_topk(q, g, 10, normalize=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-30s %7.2f ms  %6.1f M gallery rows/s  %5.0f TFLOP/s (fp32-equivalent)" % (name, dt * 1e3, G / dt / 1e6, 2.0 * Q * G * 256 / dt / 1e12), flush=True)
if ONLY:
    sys.exit(0)
a, b = res["on-the-fly split"], res["pre-split + streaming filter"]
print("indices equal:", bool(torch.equal(a[1], b[1])), " values equal:", bool(torch.equal(a[0], b[0])), " max |dv| %.1e" % float((a[0] - b[0]).abs().max()))
