"""Diagnostic (not a test): per-stage forward and per-parameter gradient error of
the HIP image encoder against the CPU oracle.  python tests/diag_visual.py [rn50|tiny] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle.fill as OF
import oracle.visual as OV
from textreid_amd.backbones.m_resnet import ModifiedResNet

tag = sys.argv[1] if len(sys.argv) > 1 else "rn50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
spec = {"rn50": OV.RN50, "tiny": OV.TINY, "rn101": OV.RN101}[tag]
seed = 2
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
m.load_state_dict(OF.fill_state(m.state_dict(), seed))
m.cuda().train()
x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed)) for k, s in OV.state_shapes(spec).items()}
for k in st:
    if OV.is_param(k): st[k].requires_grad_(True)
taps = {}
yo = OV.visual_forward(st, x, spec, True, taps)
for v in taps.values():
    if v.requires_grad: v.retain_grad()
m._debug_taps = {}
y = m(x.cuda())
names = [p for p, *_ in OV.block_plan(spec)]
for i, nme in enumerate(names):
    print("fwd %-12s %.2e" % (nme, rel(m._debug_taps[i].permute(0, 3, 1, 2), taps[nme])))
print("fwd out %.2e" % rel(y, yo))
w = OF.randn("gout:" + tag, tuple(y.shape), seed)
(yo * w).sum().backward()
m._debug_grads = []
(y * w.cuda()).sum().backward()
for i, nme in enumerate(reversed(names)):
    print('gout %-12s %.2e' % (nme, rel(m._debug_grads[i].permute(0, 3, 1, 2), taps[nme].grad)))
for k, p in m.named_parameters():
    e = rel(p.grad, st[k].grad)
    print("grad %-40s %.2e %s" % (k, e, "<<<" if e > 1e-3 else ""))
