"""GPU box: eval-mode image encoder (gallery encode, inference.py:14-26) - the P16 kernels with fused eval epilogues against the
round-2 path (BatchNorm-folded filters, on-the-fly split) and the unfolded pass; agreement of the outputs; images / s per batch size.
usage: python tools/eval_time.py [rn50|rn101] [batch sizes ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.visual as OV
from textreid_amd import ops
from textreid_amd.backbones.m_resnet import ModifiedResNet
spec = OV.RN101 if (len(sys.argv) > 1 and sys.argv[1] == "rn101") else OV.RN50
sizes = [int(a) for a in sys.argv[2:] if a.isdigit()] or [128]
ONLY_P16 = "--only-p16" in sys.argv  # (profiling runs)
torch.manual_seed(0)
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width).cuda()
with torch.no_grad():
    m.train()
    for _ in range(3):
        m(torch.randn(32, 3, 384, 128, device="cuda"))  # running statistics away from their initial values
    m.eval()
    for B in sizes:
        x = torch.randn(B, 3, 384, 128, device="cuda")
        outs = {}
        for name, fold, p16 in ((("P16 eval", True, True),) if ONLY_P16 else (("unfolded", False, False), ("folded (r02)", True, False), ("P16 eval", True, True)) * 2):
            m.fold_eval_bn, ops.USE_EVAL_P16 = fold, p16
            outs[name] = m(x); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): m(x)
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("B %4d %-13s %7.2f ms/batch (host enqueue %6.2f)  %7.0f imgs/s" % (B, name, dt * 100, th * 100, 10 * B / dt), flush=True)
        if ONLY_P16:
            continue
        ref = outs["unfolded"].double()
        for k in ("folded (r02)", "P16 eval"):
            print("   %-13s vs unfolded: max rel err %.1e" % (k, float((outs[k].double() - ref).abs().max() / ref.abs().max())))
