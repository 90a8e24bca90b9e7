#!/usr/bin/env python3
"""A/B of the main-loop schedules of trid_gemm_p16 (variant 3 = the barrier-per-tile loop; 10-13 = the software-pipelined
loops) on the RN50 layer shapes at B=128, on random and on zero-filled operands (same instruction stream, less switching
power): ms, fp32-equivalent TFLOP/s, fraction of 833, and bit-equality of the outputs with variant 3.
usage: python tools/kloop_bench.py [variants...]   (default 3 10 11 12 13)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
B = 128
WGRAD = "--wgrad" in sys.argv  # weight-gradient GEMMs instead (the main loop is chosen by TRID_WGRAD_SP: one process per form)
variants = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [3, 10, 11, 12, 13]
PEAK = 2500.0 / 3
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = {(v, z): 0.0 for v in variants for z in (0, 1)}
flops = 0.0
def run(name, M, N, K, conv, count):
    global flops
    fl = 2.0 * M * N * K
    flops += fl * count
    line = "%-26s" % name
    for zero in (0, 1):
        if conv is not None:
            H, W, Ci = conv
            x = torch.zeros(B, H, W, Ci, device=dev) if zero else torch.randn(B, H, W, Ci, device=dev).relu_()
        else:
            x = torch.zeros(M, K, device=dev) if zero else torch.randn(M, K, device=dev).relu_()
        w = torch.zeros(N, K, device=dev) if zero else torch.randn(N, K, device=dev) * 0.05
        ax, aw = ops.amax(x), ops.amax(w)
        if zero:
            ax.fill_(1.0); aw.fill_(1.0)
        xp, wp = ops.p16_pack(x, ax), ops.p16_pack(w, aw)
        ref = None
        for v in variants:
            y = torch.empty(M, N, device=dev)
            rows = ops.gemm_p16_rows(M, N, 1, v)
            st = torch.zeros((M + rows - 1) // rows, N, 4, device=dev)
            f = lambda: ops.gemm_p16(xp, wp, y, M, N, K, N, conv=conv, stats=st, variant=v, minmax=True)
            ms = t(f)
            tot[(v, zero)] += ms * count
            ok = ""
            if not zero:
                if ref is None: ref = (y.clone(), st.clone())
                else: ok = "=" if (torch.equal(y, ref[0]) and (st.shape != ref[1].shape or torch.equal(st, ref[1]))) else "DIFF(%.1e)" % float((y - ref[0]).abs().max() / ref[0].abs().max())
            line += " %s v%d %6.3f %4.0f%s" % ("z" if zero else "r", v, ms, fl / ms / 1e9, ok)
        line += " |"
    print(line, flush=True)
if WGRAD:
    import hashlib
    print("weight gradients, TRID_WGRAD_SP=%s: ms, TFLOP/s, sha1 of the result (compare across processes)" % os.environ.get("TRID_WGRAD_SP", "0"))
    tw = fw = 0.0
    def wg(name, H, W, Ci, Co, conv, count):
        global tw, fw
        Mp = B * H * W
        x, dy = torch.randn(Mp, Ci, device=dev).relu_(), torch.randn(Mp, Co, device=dev)
        if conv: x = x.view(B, H, W, Ci)
        ax, ady = ops.amax(x), ops.amax(dy)
        xp, dyp = ops.p16_pack(x, ax), ops.p16_pack(dy, ady)
        fl = 2.0 * Mp * Ci * Co * (9 if conv else 1)
        f = (lambda: ops.wgrad_p16(dyp, xp, conv=(H, W, Ci))) if conv else (lambda: ops.wgrad_p16(dyp, xp))
        g = f()
        ms = t(f)
        tw += ms * count; fw += fl * count
        mb = Mp * (Ci + Co) * 4 / 1e6  # both operands once
        print("%-28s %7.3f ms %4.0f TF  %6.1f MB %5.2f TB/s  %s" % (name, ms, fl / ms / 1e9, mb, mb / ms / 1e3, hashlib.sha1(g.cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
    torch.manual_seed(5)
    wg("l2.x 3x3 48x16 128", 48, 16, 128, 128, True, 3); wg("l3.0 3x3 48x16 256", 48, 16, 256, 256, True, 1)
    wg("l3.x 3x3 24x8 256", 24, 8, 256, 256, True, 5); wg("l4.x 3x3 24x8 512", 24, 8, 512, 512, True, 3)
    wg("l3 1x1 1024->256", 24, 8, 1024, 256, False, 5); wg("l3 1x1 256->1024", 24, 8, 256, 1024, False, 6)
    wg("l4 1x1 2048->512", 24, 8, 2048, 512, False, 2); wg("l4 1x1 512->2048", 24, 8, 512, 2048, False, 3)
    wg("l2 1x1 512->128", 48, 16, 512, 128, False, 3); wg("l2 1x1 128->512", 48, 16, 128, 512, False, 4)
    wg("l1 1x1 256->64 96x32", 96, 32, 256, 64, False, 2); wg("l1 1x1 64->256 96x32", 96, 32, 64, 256, False, 4)
    wg("l2.0 1x1 256->128 96x32", 96, 32, 256, 128, False, 1); wg("l4.ds 1x1 1024->2048", 24, 8, 1024, 2048, False, 1)
    print("total %7.3f ms  %4.0f TF  frac %.3f" % (tw, fw / tw / 1e9, fw / tw / 1e9 / PEAK))
    sys.exit(0)
print("columns: r = random operands, z = zero operands; per variant ms, TFLOP/s (fp32-equivalent), '=' bit-equal to the first variant")
shapes3 = [("l2.0 3x3 96x32 128", 96, 32, 128, 1), ("l2.x 3x3 48x16 128", 48, 16, 128, 3), ("l3.0 3x3 48x16 256", 48, 16, 256, 1),
           ("l3.x 3x3 24x8 256", 24, 8, 256, 5), ("l4.x 3x3 24x8 512", 24, 8, 512, 3)]
for name, H, W, C, cnt in shapes3:
    run(name, B * H * W, C, 9 * C, (H, W, C), cnt)
print("3x3 total (one encoder pass, 13 launches):")
for v in variants:
    print("  v%-2d random %7.3f ms %4.0f TF frac %.3f | zero %7.3f ms %4.0f TF frac %.3f" % (
        v, tot[(v, 0)], flops / tot[(v, 0)] / 1e9, flops / tot[(v, 0)] / 1e9 / PEAK, tot[(v, 1)], flops / tot[(v, 1)] / 1e9, flops / tot[(v, 1)] / 1e9 / PEAK), flush=True)
tot = {(v, z): 0.0 for v in variants for z in (0, 1)}
flops = 0.0
shapes1 = [("l2 1x1 512->128 48x16", 48 * 16, 128, 512, 3), ("l2 1x1 128->512 48x16", 48 * 16, 512, 128, 4), ("l3 1x1 1024->256 24x8", 24 * 8, 256, 1024, 5),
           ("l3 1x1 256->1024 24x8", 24 * 8, 1024, 256, 6), ("l4 1x1 2048->512 24x8", 24 * 8, 512, 2048, 2), ("l4 1x1 512->2048 24x8", 24 * 8, 2048, 512, 3),
           ("l4.ds 1x1 1024->2048", 24 * 8, 2048, 1024, 1), ("l3.0 1x1 512->256 48x16", 48 * 16, 256, 512, 1)]
for name, P, N, K, cnt in shapes1:
    run(name, B * P, N, K, None, cnt)
print("1x1 total:")
for v in variants:
    print("  v%-2d random %7.3f ms %4.0f TF frac %.3f | zero %7.3f ms %4.0f TF frac %.3f" % (
        v, tot[(v, 0)], flops / tot[(v, 0)] / 1e9, flops / tot[(v, 0)] / 1e9 / PEAK, tot[(v, 1)], flops / tot[(v, 1)] / 1e9, flops / tot[(v, 1)] / 1e9 / PEAK), flush=True)
