import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools/scratch")
import oracle.fill as OF, oracle.visual as OV
from cond_probe import fill_wc, rel
torch.set_num_threads(8)
spec = OV.RN50; B = int(sys.argv[1]); seed = 2; style = sys.argv[2]
x = OF.randn("img:probe", (B, 3, spec.height, spec.in_width), seed)
res = {}
for dt in (torch.float32, torch.float64):
    st = {}
    for k, s in OV.state_shapes(spec).items():
        if k.endswith("num_batches_tracked"): st[k] = torch.zeros((), dtype=torch.int64)
        else: st[k] = (fill_wc(k, s, seed) if style == "wc" else OF.fill(k, s, seed)).to(dt)
        if OV.is_param(k) and st[k].dtype.is_floating_point: st[k].requires_grad_(True)
    taps = {}
    y = OV.visual_forward(st, x.to(dt), spec, True, taps)
    for t in taps.values(): t.retain_grad()
    w = OF.randn("gout:probe", tuple(y.shape), seed).to(dt)
    (y * w).sum().backward()
    res[dt] = {k: (t.detach(), t.grad) for k, t in taps.items()}
for k in res[torch.float32]:
    a, ga = res[torch.float32][k]; b, gb = res[torch.float64][k]
    print("%-12s act %.1e  grad %.1e   |act| %.2e |grad| %.2e" % (k, rel(a, b), rel(ga, gb), float(b.abs().max()), float(gb.abs().max())))
