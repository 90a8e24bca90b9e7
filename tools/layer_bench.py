#!/usr/bin/env python3
"""Per-layer GEMM time of one RN50 encoder pass at B=128 (fwd / dgrad / wgrad), sequential, no overlap.
Shows where the step's MFMA time goes.  usage: python tools/layer_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
B = 128
def rnd(*s): return torch.randn(*s, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
rows = []
FILT = os.environ.get("TRID_LB_FILTER", "")
PREC = int(os.environ.get("TRID_LB_PREC", "6"))  # 6 = bf16x6, 16 = fp16x3 (operand amax passes NOT included: precomputed)
def am(t): return ops.amax(t) if PREC == 16 else None
def conv1(name, P, Ci, Co, count=1, acc=False):
    if FILT and not any(f in name for f in FILT.split(",")): return
    M = B * P
    x, w, dy = rnd(M, Ci), rnd(Co, Ci), rnd(M, Co)
    dx = torch.empty(M, Ci, device=dev)
    fl = 2.0 * M * Ci * Co
    ax, aw, ady = am(x), am(w), am(dy)
    rows.append((name + " fwd", count, t(lambda: ops.conv1x1(x, w, stats=True, prec=PREC, aa=ax, ba=aw)), fl))
    if os.environ.get("TRID_DGRAD_T"):  # data gradient against a pre-transposed weight copy (K-contiguous B)
        wT = w.t().contiguous()
        awT = am(wT)
        rows.append((name + " dgrad", count, t(lambda: ops.linear(dy, wT, out=dx, accumulate=acc, prec=PREC, aa=ady, ba=awT)), fl))
    else:
        rows.append((name + " dgrad", count, t(lambda: ops.matmul_nn(dy, w, out=dx, accumulate=acc, prec=PREC, aa=ady, ba=aw)), fl))
    rows.append((name + " wgrad", count, t(lambda: ops.conv1x1_wgrad(dy, x, prec=PREC, aa=ady, ba=ax)), fl))
def conv3(name, H, W, Ci, Co, count=1):
    if FILT and not any(f in name for f in FILT.split(",")): return
    x, w, dy = rnd(B, H, W, Ci), rnd(Co, 9 * Ci), rnd(B, H, W, Co)
    wt = rnd(Ci, 9 * Co)
    fl = 2.0 * B * H * W * Ci * Co * 9
    ax, aw, ady, awt = am(x), am(w), am(dy), am(wt)
    rows.append((name + " fwd", count, t(lambda: ops.conv3x3(x, w, stats=True, prec=PREC, aa=ax, ba=aw)), fl))
    rows.append((name + " dgrad", count, t(lambda: ops.conv3x3(dy, wt, prec=PREC, aa=ady, ba=awt)), fl))
    rows.append((name + " wgrad", count, t(lambda: ops.conv3x3_wgrad(dy, x, prec=PREC, aa=ady, ba=ax)), fl))
conv3("stem.conv2 3x3 32->32 @192x64", 192, 64, 32, 32)
conv3("stem.conv3 3x3 32->64 @192x64", 192, 64, 32, 64)
# layer1 @96x32 (3072 px)
conv1("l1.0.conv1 64->64", 3072, 64, 64); conv3("l1.x.conv2 3x3 64 @96x32", 96, 32, 64, 64, 3)
conv1("l1.x.conv3 64->256", 3072, 64, 256, 3); conv1("l1.0.down 64->256", 3072, 64, 256)
conv1("l1.1-2.conv1 256->64", 3072, 256, 64, 2, acc=True)
# layer2
conv1("l2.0.conv1 256->128 @3072", 3072, 256, 128, acc=True); conv3("l2.0.conv2 3x3 128 @96x32", 96, 32, 128, 128)
conv1("l2.x.conv3 128->512 @768", 768, 128, 512, 4); conv1("l2.0.down 256->512 @768", 768, 256, 512)
conv1("l2.1-3.conv1 512->128", 768, 512, 128, 3, acc=True); conv3("l2.1-3.conv2 3x3 128 @48x16", 48, 16, 128, 128, 3)
# layer3
conv1("l3.0.conv1 512->256 @768", 768, 512, 256, acc=True); conv3("l3.0.conv2 3x3 256 @48x16", 48, 16, 256, 256)
conv1("l3.x.conv3 256->1024 @192", 192, 256, 1024, 6); conv1("l3.0.down 512->1024 @192", 192, 512, 1024)
conv1("l3.1-5.conv1 1024->256", 192, 1024, 256, 5, acc=True); conv3("l3.1-5.conv2 3x3 256 @24x8", 24, 8, 256, 256, 5)
# layer4
conv1("l4.0.conv1 1024->512", 192, 1024, 512, acc=True); conv3("l4.x.conv2 3x3 512 @24x8", 24, 8, 512, 512, 3)
conv1("l4.x.conv3 512->2048", 192, 512, 2048, 3); conv1("l4.0.down 1024->2048", 192, 1024, 2048)
conv1("l4.1-2.conv1 2048->512", 192, 2048, 512, 2, acc=True)
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print("%-40s %5s %9s %9s %8s" % ("layer", "count", "ms/call", "ms total", "TF/s"))
for name, c, ms, fl in rows:
    print("%-40s %5d %9.3f %9.3f %8.1f" % (name, c, ms, ms * c, fl / ms / 1e9))
    tot[name.split()[-1]] += ms * c
print("totals per encoder pass: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms -> step (2 fwd + dgrad + wgrad) = %.2f ms" % (tot["fwd"], tot["dgrad"], tot["wgrad"], 2 * tot["fwd"] + tot["dgrad"] + tot["wgrad"]))
