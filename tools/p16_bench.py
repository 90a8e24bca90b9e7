#!/usr/bin/env python3
"""A/B of the pre-split (P16, LDS-DMA) GEMM against the on-the-fly fp16-split kernel on the RN50 layer shapes at B=128:
forward 3x3 / 1x1 convolutions with the BatchNorm-partials epilogue.  Checks the results against each other first.
usage: python tools/p16_bench.py [variants...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
B = 128
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
print("%-34s %9s %s" % ("layer", "old ms/TF", " ".join("v%d ms/TF (err)    " % v for v in variants)))
tot_old, tot = 0.0, {v: 0.0 for v in variants}
def conv3(name, H, W, Ci, Co, count=1):
    global tot_old
    x, w = torch.randn(B, H, W, Ci, device=dev).relu_(), torch.randn(Co, 9 * Ci, device=dev) * 0.05
    ax, aw = ops.amax(x), ops.amax(w)
    fl = 2.0 * B * H * W * Ci * Co * 9
    y0, st0 = ops.conv3x3(x, w, stats=True, prec=16, aa=ax, ba=aw)
    ms0 = t(lambda: ops.conv3x3(x, w, stats=True, prec=16, aa=ax, ba=aw))
    tot_old += ms0 * count
    xp, wp = ops.p16_pack(x, ax), ops.p16_pack(w, aw)
    M = B * H * W
    out = "%-34s %5.3f/%3.0f" % (name, ms0, fl / ms0 / 1e9)
    for v in variants:
        y = torch.empty(B, H, W, Co, device=dev); st = ops.stats_buffer(M, Co, x)
        f = lambda: ops.gemm_p16(xp, wp, y, M, Co, 9 * Ci, Co, conv=(H, W, Ci), stats=st, variant=v)
        ms = t(f)
        tot[v] += ms * count
        out += "  %5.3f/%3.0f (%.0e,%.0e)" % (ms, fl / ms / 1e9, rel(y, y0), rel(st, st0))
    print(out, flush=True)
def conv1(name, P, Ci, Co, count=1):
    global tot_old
    M = B * P
    x, w = torch.randn(M, Ci, device=dev).relu_(), torch.randn(Co, Ci, device=dev) * 0.05
    ax, aw = ops.amax(x), ops.amax(w)
    fl = 2.0 * M * Ci * Co
    y0, st0 = ops.conv1x1(x, w, stats=True, prec=16, aa=ax, ba=aw)
    ms0 = t(lambda: ops.conv1x1(x, w, stats=True, prec=16, aa=ax, ba=aw))
    tot_old += ms0 * count
    xp, wp = ops.p16_pack(x, ax), ops.p16_pack(w, aw)
    out = "%-34s %5.3f/%3.0f" % (name, ms0, fl / ms0 / 1e9)
    for v in variants:
        y = torch.empty(M, Co, device=dev); st = ops.stats_buffer(M, Co, x)
        f = lambda: ops.gemm_p16(xp, wp, y, M, Co, Ci, Co, stats=st, variant=v)
        ms = t(f)
        tot[v] += ms * count
        out += "  %5.3f/%3.0f (%.0e,%.0e)" % (ms, fl / ms / 1e9, rel(y, y0), rel(st, st0))
    print(out, flush=True)
def wg3(name, H, W, Ci, Co, count=1):
    x, dy = torch.randn(B, H, W, Ci, device=dev).relu_(), torch.randn(B, H, W, Co, device=dev)
    ax, ady = ops.amax(x), ops.amax(dy)
    fl = 2.0 * B * H * W * Ci * Co * 9
    g0 = ops.conv3x3_wgrad(dy, x, prec=16, aa=ady, ba=ax)
    ms0 = t(lambda: ops.conv3x3_wgrad(dy, x, prec=16, aa=ady, ba=ax))
    xp, dyp = ops.p16_pack(x, ax), ops.p16_pack(dy, ady)
    g1 = ops.wgrad_p16(dyp, xp, conv=(H, W, Ci))
    ms1 = t(lambda: ops.wgrad_p16(dyp, xp, conv=(H, W, Ci)))
    print("wgrad %-28s old %6.3f ms %4.0f TF | p16 %6.3f ms %4.0f TF  (err %.0e)" % (name, ms0, fl / ms0 / 1e9, ms1, fl / ms1 / 1e9, rel(g1, g0)), flush=True)
    return ms0 * count, ms1 * count
def wg1(name, P, Ci, Co, count=1):
    M = B * P
    x, dy = torch.randn(M, Ci, device=dev).relu_(), torch.randn(M, Co, device=dev)
    ax, ady = ops.amax(x), ops.amax(dy)
    fl = 2.0 * M * Ci * Co
    g0 = ops.conv1x1_wgrad(dy, x, prec=16, aa=ady, ba=ax)
    ms0 = t(lambda: ops.conv1x1_wgrad(dy, x, prec=16, aa=ady, ba=ax))
    xp, dyp = ops.p16_pack(x, ax), ops.p16_pack(dy, ady)
    g1 = ops.wgrad_p16(dyp, xp)
    ms1 = t(lambda: ops.wgrad_p16(dyp, xp))
    print("wgrad %-28s old %6.3f ms %4.0f TF | p16 %6.3f ms %4.0f TF  (err %.0e)" % (name, ms0, fl / ms0 / 1e9, ms1, fl / ms1 / 1e9, rel(g1, g0)), flush=True)
    return ms0 * count, ms1 * count
if os.environ.get("P16_WGRAD", "1") == "1":
    tw = [0.0, 0.0]
    for r in (wg3("l1 3x3 64 @96x32", 96, 32, 64, 64, 3), wg1("l1 conv3 64->256", 3072, 64, 256, 4), wg1("l1 conv1 256->64", 3072, 256, 64, 2),
              wg1("l2.0 conv1 256->128", 3072, 256, 128), wg3("l2.0 3x3 128 @96x32", 96, 32, 128, 128), wg1("l2 conv3 128->512", 768, 128, 512, 4),
              wg1("l2.0 down 256->512", 768, 256, 512), wg1("l2 conv1 512->128", 768, 512, 128, 3), wg3("l2 3x3 128 @48x16", 48, 16, 128, 128, 3),
              wg1("l3.0 conv1 512->256", 768, 512, 256), wg3("l3.0 3x3 256 @48x16", 48, 16, 256, 256), wg1("l3 conv3 256->1024", 192, 256, 1024, 6),
              wg1("l3.0 down 512->1024", 192, 512, 1024), wg1("l3 conv1 1024->256", 192, 1024, 256, 5), wg3("l3 3x3 256 @24x8", 24, 8, 256, 256, 5),
              wg1("l4.0 conv1 1024->512", 192, 1024, 512), wg3("l4 3x3 512 @24x8", 24, 8, 512, 512, 3), wg1("l4 conv3 512->2048", 192, 512, 2048, 3),
              wg1("l4.0 down 1024->2048", 192, 1024, 2048), wg1("l4 conv1 2048->512", 192, 2048, 512, 2)):
        tw[0] += r[0]; tw[1] += r[1]
    print("weight-gradient GEMM total per encoder pass: old %.2f ms, p16 %.2f ms" % tuple(tw))
if os.environ.get("P16_FWD", "1") != "1":
    sys.exit(0)
# pack / unpack round trip
z = torch.randn(1000, 96, device=dev) * 3
zp = ops.p16_pack(z)
print("pack/unpack round trip rel err %.1e" % rel(zp.unpack(), z))
conv3("stem.conv2 3x3 32->32 @192x64", 192, 64, 32, 32)
conv3("stem.conv3 3x3 32->64 @192x64", 192, 64, 32, 64)
conv1("l1.0.conv1 64->64", 3072, 64, 64); conv3("l1.x.conv2 3x3 64 @96x32", 96, 32, 64, 64, 3)
conv1("l1.x.conv3 64->256", 3072, 64, 256, 4); conv1("l1.1-2.conv1 256->64", 3072, 256, 64, 2)
conv1("l2.0.conv1 256->128 @3072", 3072, 256, 128); conv3("l2.0.conv2 3x3 128 @96x32", 96, 32, 128, 128)
conv1("l2.x.conv3 128->512 @768", 768, 128, 512, 4); conv1("l2.0.down 256->512 @768", 768, 256, 512)
conv1("l2.1-3.conv1 512->128", 768, 512, 128, 3); conv3("l2.1-3.conv2 3x3 128 @48x16", 48, 16, 128, 128, 3)
conv1("l3.0.conv1 512->256 @768", 768, 512, 256); conv3("l3.0.conv2 3x3 256 @48x16", 48, 16, 256, 256)
conv1("l3.x.conv3 256->1024 @192", 192, 256, 1024, 6); conv1("l3.0.down 512->1024 @192", 192, 512, 1024)
conv1("l3.1-5.conv1 1024->256", 192, 1024, 256, 5); conv3("l3.1-5.conv2 3x3 256 @24x8", 24, 8, 256, 256, 5)
conv1("l4.0.conv1 1024->512", 192, 1024, 512); conv3("l4.x.conv2 3x3 512 @24x8", 24, 8, 512, 512, 3)
conv1("l4.x.conv3 512->2048", 192, 512, 2048, 3); conv1("l4.0.down 1024->2048", 192, 1024, 2048)
conv1("l4.1-2.conv1 2048->512", 192, 2048, 512, 2)
print("forward GEMM total per encoder pass: old %.2f ms; " % tot_old + "  ".join("v%d %.2f ms" % (v, tot[v]) for v in variants))
