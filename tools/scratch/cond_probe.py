"""Probe: how well-conditioned is a random-weight RN50/RN101 step under a given fill style?
err(fp32 oracle vs fp64 oracle) per output/gradient."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import oracle.fill as OF
import oracle.visual as OV
torch.set_num_threads(8)

def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def fill_wc(name, shape, seed, g3=(0.15, 0.35), g12=(0.8, 1.2)):
    rs = OF._rs(name, seed)
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    parts = name.split(".")
    leaf = parts[-1]
    if leaf == "weight" and len(shape) == 1:
        lo, hi = g3 if (len(parts) >= 2 and parts[-2] == "bn3" and parts[0].startswith("layer")) else g12
        a = rs.uniform(lo, hi, size=n)
        return torch.from_numpy(a.astype(np.float32).reshape(shape))
    return OF.fill(name, shape, seed)

def run(spec, B, seed, style, **kw):
    x = OF.randn("img:probe", (B, 3, spec.height, spec.in_width), seed)
    res = {}
    for dt in (torch.float32, torch.float64):
        st = {}
        for k, s in OV.state_shapes(spec).items():
            if k.endswith("num_batches_tracked"):
                st[k] = torch.zeros((), dtype=torch.int64)
            else:
                st[k] = (fill_wc(k, s, seed, **kw) if style == "wc" else OF.fill(k, s, seed)).to(dt)
            if OV.is_param(k) and st[k].dtype.is_floating_point:
                st[k].requires_grad_(True)
        t0 = time.time()
        y = OV.visual_forward(st, x.to(dt), spec, True)
        w = OF.randn("gout:probe", tuple(y.shape), seed).to(dt)
        (y * w).sum().backward()
        with torch.no_grad():
            ye = OV.visual_forward(st, x.to(dt), spec, False)
        res[dt] = (y.detach(), {k: v.grad for k, v in st.items() if v.requires_grad}, ye)
        print("  %s: %.1fs" % (dt, time.time() - t0), flush=True)
    y32, g32, e32 = res[torch.float32]
    y64, g64, e64 = res[torch.float64]
    errs = {k: rel(g32[k], g64[k]) for k in g32}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("style=%s B=%d %s: out %.1e  eval_cold %.1e  |y| %.2e |ye| %.2e  median grad err %.1e  worst %s" % (
        style, B, kw, rel(y32, y64), rel(e32, e64), float(y64.abs().max()), float(e64.abs().max()),
        float(np.median(list(errs.values()))), [(k, "%.1e" % v) for k, v in worst]), flush=True)

if __name__ == "__main__":
    spec = {"rn50": OV.RN50, "rn101": OV.RN101, "tiny": OV.TINY}[sys.argv[1]]
    B = int(sys.argv[2])
    for style in sys.argv[3:]:
        if style.startswith("wc"):
            parts = style.split(":")
            kw = {}
            if len(parts) > 1:
                lo, hi = (float(v) for v in parts[1].split(","))
                kw["g3"] = (lo, hi)
            run(spec, B, 2, "wc", **kw)
        else:
            run(spec, B, 2, "he")
