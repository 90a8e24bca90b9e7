"""GPU box: how loose are the analytic output bounds of the eval-mode epilogues (csrc/gemm_common.h EvalBound)?  Per fused layer of
an eval pass: bound / true maximum as a power of two (the bits the two-plane fp16 format has to spare).  Default-initialised
weights with warmed running statistics, and the golden fixture's He-style weights.  usage: python tools/eval_bounds.py [rn50|rn101]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.fill as OF
import oracle.visual as OV
from textreid_amd.backbones.m_resnet import ModifiedResNet
spec = OV.RN101 if "rn101" in sys.argv else OV.RN50
for label in ("default init", "He-style fill (oracle.fill)"):
    torch.manual_seed(0)
    m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
    if label.startswith("He"):
        m.load_state_dict(OF.fill_state(m.state_dict(), 5, "", style="he"))
    m.cuda()
    with torch.no_grad():
        m.train()
        for _ in range(5):
            m(torch.randn(32, 3, 384, 128, device="cuda"))
        m.eval()
        m._debug_eval_bounds = []
        m(torch.randn(128, 3, 384, 128, device="cuda"))
        torch.cuda.synchronize()
    bits = [(n, math.log2(float(t.amax) / max(float(t.tmax), 1e-30))) for n, t in m._debug_eval_bounds]
    m._debug_eval_bounds = None
    b = sorted(v for _, v in bits)
    print("%s, %s: %d fused epilogues; log2(bound / true maximum): min %.1f  median %.1f  max %.1f" % (label, "RN101" if spec is OV.RN101 else "RN50", len(b), b[0], b[len(b) // 2], b[-1]))
    print("   loosest:", ", ".join("%s %.1f" % (n, v) for n, v in sorted(bits, key=lambda kv: -kv[1])[:5]))
    print("   tightest:", ", ".join("%s %.1f" % (n, v) for n, v in sorted(bits, key=lambda kv: kv[1])[:3]))
