import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import oracle.fill as OF, oracle.visual as OV
torch.set_num_threads(8)
def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def chan_sign(tag, n, seed):
    return np.where(OF._rs("sign:" + tag, seed).uniform(size=n) < 0.5, -1.0, 1.0)

def fill_margin(name, shape, seed, margin=4.0):
    """BatchNorm affine chosen so that every ReLU decision has a margin of `margin` sigma."""
    parts = name.split(".")
    leaf = parts[-1]
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1 and leaf in ("weight", "bias") and (parts[-2].startswith("bn") or parts[-2] == "1"):
        n = shape[0]
        bn = ".".join(parts[:-1])
        resid = parts[0].startswith("layer") and (parts[-2] == "bn3" or parts[-2] == "1")
        g = OF._rs(bn + ".weight", seed).uniform(0.2, 0.4, size=n) if resid else OF._rs(bn + ".weight", seed).uniform(0.8, 1.2, size=n)
        if leaf == "weight":
            a = g
        else:
            # block outputs share one sign pattern per residual layer (the identity path keeps channel identity)
            tag = parts[0] if resid else bn
            a = margin * g * chan_sign(tag, n, seed)
        return torch.from_numpy(np.asarray(a, dtype=np.float32))
    return OF.fill(name, shape, seed)

spec = {"rn50": OV.RN50, "rn101": OV.RN101, "tiny": OV.TINY}[sys.argv[1]]; B = int(sys.argv[2]); seed = 2; margin = float(sys.argv[3])
x = OF.randn("img:probe", (B, 3, spec.height, spec.in_width), seed)
res = {}
for dt in (torch.float32, torch.float64):
    st = {}
    for k, s in OV.state_shapes(spec).items():
        if k.endswith("num_batches_tracked"): st[k] = torch.zeros((), dtype=torch.int64)
        else: st[k] = fill_margin(k, s, seed, margin).to(dt)
        if OV.is_param(k) and st[k].dtype.is_floating_point: st[k].requires_grad_(True)
    taps = {}
    t0 = time.time()
    y = OV.visual_forward(st, x.to(dt), spec, True, taps)
    for t in taps.values(): t.retain_grad()
    w = OF.randn("gout:probe", tuple(y.shape), seed).to(dt)
    (y * w).sum().backward()
    print(dt, "%.1fs" % (time.time() - t0), flush=True)
    res[dt] = ({k: (t.detach(), t.grad) for k, t in taps.items()}, {k: v.grad for k, v in st.items() if v.requires_grad})
for k in res[torch.float32][0]:
    a, ga = res[torch.float32][0][k]; b, gb = res[torch.float64][0][k]
    print("%-12s act %.1e  grad %.1e   |act| %.2e |grad| %.2e  frac>0 %.3f" % (k, rel(a, b), rel(ga, gb), float(b.abs().max()), float(gb.abs().max()), float((b > 0).double().mean())))
g32, g64 = res[torch.float32][1], res[torch.float64][1]
errs = {k: rel(g32[k], g64[k]) for k in g32 if not k.endswith("k_proj.bias")}
print("param grads: median %.1e max %.1e" % (float(np.median(list(errs.values()))), max(errs.values())), sorted(errs.items(), key=lambda kv: -kv[1])[:5])
# cold eval
ev = {}
for dt in (torch.float32, torch.float64):
    st = {}
    for k, s in OV.state_shapes(spec).items():
        if k.endswith("num_batches_tracked"): st[k] = torch.zeros((), dtype=torch.int64)
        else: st[k] = fill_margin(k, s, seed, margin).to(dt)
    with torch.no_grad():
        ev[dt] = OV.visual_forward(st, x.to(dt), spec, False)
print("cold eval: err %.1e |y| %.2e" % (rel(ev[torch.float32], ev[torch.float64]), float(ev[torch.float64].abs().max())))
