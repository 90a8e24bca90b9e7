import os, sys
import torch as T
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
T.manual_seed(0)
M_, N, K = 256, 256, 64
x, w = T.relu(T.randn(M_, K)), T.randn(N, K) * 0.2
ident = T.relu(T.randn(M_, N))
gamma, beta = T.randn(N).abs() + 0.5, T.randn(N)
d = lambda t: t.cuda()
xp, wp, ip = ops.p16_pack(d(x)), ops.p16_pack(d(w)), ops.p16_pack(d(ident))
y_ref, st_ref = ops.conv_p16(xp, wp)
b_ref = ops.amax_slot(y_ref.device)
fin_ref = ops.bn_finalize_minmax(st_ref, M_, d(gamma), d(beta), None, None, False, b_ref)
out_ref, mask_ref = ops.bn_apply_p16(y_ref, fin_ref, b_ref, relu=True, res=ip, bound_res=ip.amax, want_mask=True)
st = ops.conv1x1_stats_p16(xp, wp)
b = ops.amax_slot(y_ref.device)
fin = ops.bn_finalize_minmax(st, M_, d(gamma), d(beta), None, None, False, b)
out, mask, y = ops.conv1x1_bn_res_p16(xp, wp, fin, b, ip, relu=True, want_mask=True, keep_y=True)
print("y equal", T.equal(y, y_ref), "amax", float(out.amax), float(out_ref.amax))
a, r = out.unpack().cpu(), out_ref.unpack().cpu()
bad = ((a - r).abs() > 1e-6).nonzero()
print("bad", bad.shape[0], "of", a.numel(), "max diff", float((a - r).abs().max()))
if bad.shape[0]:
    print(" rows", bad[:, 0].unique().tolist()[:40]); print(" cols", bad[:, 1].unique().tolist()[:70])
    for i in range(6):
        rr, cc = bad[i].tolist(); print(rr, cc, float(a[rr, cc]), float(r[rr, cc]))
raw = (out.data.view(T.int32) != out_ref.data.view(T.int32)).nonzero()
print("raw dword mismatches", raw.shape[0], raw[:8].tolist())
mm = (mask != mask_ref).nonzero()
print("mask word mismatches", mm.shape[0], "of", mask.numel(), mm[:8].flatten().tolist())
if mm.shape[0]:
    i = int(mm[0]); print(hex(int(mask[i]) & (2**64-1)), hex(int(mask_ref[i]) & (2**64-1)))
