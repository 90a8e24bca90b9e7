"""Diagnostic (GPU box): HIP vs CPU-oracle gradients w.r.t. every block output of a full-size encoder;
localises ReLU-mask flips (isolated elements with an O(1) relative deviation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle.fill as OF, oracle.visual as OV
from textreid_amd.backbones.m_resnet import ModifiedResNet

tag = sys.argv[1] if len(sys.argv) > 1 else "rn50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spec = {"rn50": OV.RN50, "rn101": OV.RN101}[tag]
dev = torch.device("cuda")
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
m.load_state_dict(OF.fill_state(m.state_dict(), seed, style="margin"))
m.to(dev).train()
x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
m._debug_taps = {}
m._debug_grads = []
y = m(x.to(dev))
w = OF.randn("gout:" + tag, tuple(y.shape), seed)
(y * w.to(dev)).sum().backward()
st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed, style="margin")) for k, s in OV.state_shapes(spec).items()}
for k in st:
    if OV.is_param(k):
        st[k].requires_grad_(True)
taps = {}
yo = OV.visual_forward(st, x, spec, True, taps)
for t in (v for v in taps.values() if torch.is_tensor(v)):
    t.retain_grad()
(yo * w).sum().backward()
names = [k for k in taps if k.startswith("layer")]
print("oracle relu_min %.2e" % taps["relu_min"])
hip_g = dict(zip(reversed(names), m._debug_grads))
for k in names:
    a = m._debug_taps[k].permute(0, 3, 1, 2).cpu()
    ga = hip_g[k].permute(0, 3, 1, 2).cpu()
    b, gb = taps[k].detach(), taps[k].grad
    da = (a - b).abs()
    dg = (ga - gb).abs()
    big = dg > 1e-2 * gb.abs().max()
    print("%-10s act err %.1e  grad err %.1e  n(|dgrad|>1e-2 max) %d  maskdiff %d" % (
        k, float(da.max() / b.abs().max()), float(dg.max() / gb.abs().max()), int(big.sum()), int(((a > 0) != (b > 0)).sum())))
    if int(big.sum()) and int(big.sum()) < 20:
        idx = big.nonzero()
        for i in idx[:5]:
            i = tuple(int(v) for v in i)
            print("     at", i, "hip grad %.4e ora %.4e  act hip %.4e ora %.4e" % (float(ga[i]), float(gb[i]), float(a[i]), float(b[i])))
