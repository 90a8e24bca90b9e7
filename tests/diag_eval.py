"""Diagnostic (not a test): eval-mode per-stage error of the HIP image encoder vs the fp64 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle.fill as OF
import oracle.visual as OV
from textreid_amd.backbones.m_resnet import ModifiedResNet
tag = sys.argv[1] if len(sys.argv) > 1 else "rn101"
spec = {"rn50": OV.RN50, "tiny": OV.TINY, "rn101": OV.RN101}[tag]
seed, B = 2, 2
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
m.load_state_dict(OF.fill_state(m.state_dict(), seed)); m.cuda()
x = OF.randn("img:" + tag, (B, 3, spec.height, spec.in_width), seed)
for dt in (torch.float64, torch.float32):
    st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed).to(dt)) for k, s in OV.state_shapes(spec).items()}
    with torch.no_grad():
        OV.visual_forward(st, x.to(dt), spec, True)
        taps = {}
        yo = OV.visual_forward(st, x.to(dt), spec, False, taps)
    if dt == torch.float64: t64, y64 = taps, yo
    else: t32, y32 = taps, yo
with torch.no_grad():
    m.train(); m(x.cuda()); m.eval(); m._debug_taps = {}
    y = m(x.cuda())
names = [p for p, *_ in OV.block_plan(spec)]
for i, nme in enumerate(names):
    print("eval %-12s hip %.2e  cpu32 %.2e  |max| %.3g" % (nme, rel(m._debug_taps[i].permute(0, 3, 1, 2), t64[nme]), rel(t32[nme], t64[nme]), float(t64[nme].abs().max())))
print("eval out hip %.2e cpu32 %.2e" % (rel(y, y64), rel(y32, y64)))
