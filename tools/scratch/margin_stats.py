import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import oracle.fill as OF, oracle.visual as OV
import torch.nn.functional as F
torch.set_num_threads(8)
margin = float(sys.argv[1]); B = int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2
OF.MARGIN = margin
spec = OV.RN50
cnt = {1e-2: 0, 1e-3: 0, 1e-4: 0, 1e-5: 0}; tot = 0; mn = 1e9; wrong = 0
orig = F.relu
def relu(x, *a, **k):
    global tot, mn, wrong
    ax = x.detach().abs()
    tot += ax.numel(); mn = min(mn, float(ax.min()))
    for t in cnt: cnt[t] += int((ax < t).sum())
    return orig(x, *a, **k)
OV.F.relu = relu
st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, s, seed, style="margin").double()) for k, s in OV.state_shapes(spec).items()}
x = OF.randn("img:rn50", (B, 3, spec.height, spec.in_width), seed).double()
with torch.no_grad():
    OV.visual_forward(st, x, spec, True)
print("margin", margin, "B", B, "relu inputs", tot, "min|y| %.2e" % mn, {k: v for k, v in cnt.items()})
