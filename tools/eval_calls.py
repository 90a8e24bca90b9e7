#!/usr/bin/env python3
"""GPU box: every kernel-library call of ONE eval-mode image-encoder pass (gallery encode, B images) with its scalar arguments and
its event time, grouped by (entry point, arguments) and sorted by total time; the same for the text encoder with --text.
usage: python tools/eval_calls.py [rn50|rn101] [B] [--order]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import oracle.visual as OV  # noqa: E402
from textreid_amd import lib  # noqa: E402
from textreid_amd.backbones.m_resnet import ModifiedResNet  # noqa: E402

spec = OV.RN101 if "rn101" in sys.argv else OV.RN50
B = next((int(a) for a in sys.argv[1:] if a.isdigit()), 128)
torch.manual_seed(0)
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width).cuda().eval()
x = torch.randn(B, 3, 384, 128, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    lib.TRACE = []
    m(x)
    torch.cuda.synchronize()
tr, lib.TRACE = lib.TRACE, None
if "--order" in sys.argv:
    for name, scal, e0, e1 in tr:
        print("%8.1f us  %s %s" % (e0.elapsed_time(e1) * 1e3, name.replace("trid_", ""), " ".join(str(a) for a in scal)))
c = collections.OrderedDict()
for name, scal, e0, e1 in tr:
    v = c.setdefault((name, scal), [0, 0.0])
    v[0] += 1
    v[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in c.values())
print("%d calls, %.2f ms of event time for %d images" % (len(tr), tot / 1e3, B))
kinds = collections.defaultdict(float)
for (name, scal), (cnt, us) in c.items():
    kinds[name] += us
for name, us in sorted(kinds.items(), key=lambda kv: -kv[1]):
    print("  %8.1f us  %s" % (us, name))
for (name, scal), (cnt, us) in sorted(c.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  x%-3d avg %7.1f  %s %s" % (us, cnt, us / cnt, name.replace("trid_", ""), " ".join(str(a) for a in scal)))
