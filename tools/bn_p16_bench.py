#!/usr/bin/env python3
"""The BatchNorm backward passes of the P16 data flow (reduce + fold + apply: ops.bn_bwd_p16) on the RN50 layer shapes at B=128,
isolated: ms per call and TB/s of the algorithmic streams (reduce: g + y; apply: g + y + dy), plus a digest of the results
(compare across builds: TRID_LIB_PATH selects the library).  usage: python tools/bn_p16_bench.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreid_amd import ops
dev = torch.device("cuda")
B = 128
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
shapes = [("l1 bn3 96x32x256", 96, 32, 256, 3), ("l2 bn3 48x16x512", 48, 16, 512, 4), ("l3 bn3 24x8x1024", 24, 8, 1024, 6), ("l4 bn3 24x8x2048", 24, 8, 2048, 3),
          ("l2 bn2 48x16x128", 48, 16, 128, 4), ("l3 bn2 24x8x256", 24, 8, 256, 6), ("l4 bn2 24x8x512", 24, 8, 512, 3), ("l1 bn2 96x32x64", 96, 32, 64, 3)]
torch.manual_seed(3)
tot = 0.0
h = hashlib.sha1()
print("library:", os.environ.get("TRID_LIB_PATH", "default"))
for name, H, W, C, cnt in shapes:
    y = torch.randn(B, H, W, C, device=dev); g = torch.randn_like(y)
    st = ops.BNState(C, y)
    st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.uniform_(0.5, 1.5); st.shift.normal_()
    bound = ops.amax_slot(dev); bound.fill_(8.0)
    out, mask = ops.bn_apply_p16(y, st, bound, relu=True, want_mask=True)
    mb = y.numel() * 4 / 1e6
    mode3 = "bn3" in name
    f = (lambda: ops.bn_bwd_p16(g, y, st, 3, act=mask)) if mode3 else (lambda: ops.bn_bwd_p16(g, y, st, 1))
    dy, dg, db, _ = f()
    torch.cuda.synchronize()
    for x in (dy.data, dg, db): h.update(x.cpu().numpy().tobytes())
    ms = t(f)
    tot += ms * cnt
    print("%-22s %7.1f MB  %7.3f ms  %5.2f TB/s (5 streams: reduce g+y, apply g+y+dy)" % (name, mb, ms, 5 * mb / ms / 1e3), flush=True)
print("backward: weighted total %.3f ms   digest %s" % (tot, h.hexdigest()[:16]))
# forward: act(bn(y)) -> P16 (2 streams), and + identity residual (P16) with the ReLU bit mask (3 streams)
tot = 0.0
h = hashlib.sha1()
for name, H, W, C, cnt, with_res in [("l2 bn1 96x32x128", 96, 32, 128, 2, False), ("l3 bn2 24x8x256", 24, 8, 256, 24, False), ("l4 bn2 24x8x512", 24, 8, 512, 12, False),
                                     ("l2 bn2 48x16x128", 48, 16, 128, 14, False), ("l1 bn3+id 96x32x256", 96, 32, 256, 4, True), ("l2 bn3+id 48x16x512", 48, 16, 512, 6, True),
                                     ("l3 bn3+id 24x8x1024", 24, 8, 1024, 10, True), ("l4 bn3+id 24x8x2048", 24, 8, 2048, 4, True)]:
    y = torch.randn(B, H, W, C, device=dev)
    st = ops.BNState(C, y)
    st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.uniform_(0.5, 1.5); st.shift.normal_()
    bound = ops.amax_slot(dev); bound.fill_(8.0)
    mb = y.numel() * 4 / 1e6
    if with_res:
        r = torch.randn_like(y); ra = ops.amax(r); rp = ops.p16_pack(r, ra)
        f = lambda: ops.bn_apply_p16(y, st, bound, relu=True, res=rp, bound_res=ra, want_mask=True)
    else:
        f = lambda: ops.bn_apply_p16(y, st, bound, relu=True)
    o = f()
    torch.cuda.synchronize()
    h.update((o[0] if with_res else o).data.cpu().numpy().tobytes())
    ms = t(f)
    tot += ms * cnt
    print("%-22s %7.1f MB  %7.3f ms  %5.2f TB/s (%d streams)" % (name, mb, ms, (3 if with_res else 2) * mb / ms / 1e3, 3 if with_res else 2), flush=True)
print("forward: weighted total %.3f ms   digest %s" % (tot, h.hexdigest()[:16]))
