#!/usr/bin/env python3
"""GPU box: every kernel-library call of ONE eager train step on ONE stream (TRID_SERIAL=1 is set here), with its scalar
arguments (shapes) and its event time; grouped by (entry point, arguments), sorted by total time.  The durations are
un-overlapped: what each call costs with the GPU to itself.
usage: python tools/step_calls.py [min_total_us]"""
import collections
import os
import sys

os.environ["TRID_SERIAL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from textreid_amd import lib  # noqa: E402
from textreid_amd.caption import CaptionBatch  # noqa: E402
from textreid_amd.config import moco_cfg  # noqa: E402
from textreid_amd.model import build_model  # noqa: E402
from textreid_amd.solver import make_optimizer  # noqa: E402

floor = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = 128
cfg = moco_cfg("m_resnet50", K=8192)
model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(dev)
model.train()
opt = make_optimizer(cfg, model)
batches = [bench.synth_batch(B, s, dev, 1234) for s in range(2)]


def step(i):
    images, tokens, lengths, ids = batches[i % 2]
    ld = model(images, CaptionBatch(tokens, lengths, ids % 11003, max_len=64))
    opt.zero_grad()
    sum(ld.values()).backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
lib.TRACE = []
step(3)
torch.cuda.synchronize()
tr, lib.TRACE = lib.TRACE, None
c = collections.OrderedDict()
for name, scal, e0, e1 in tr:
    k = (name, scal)
    v = c.setdefault(k, [0, 0.0])
    v[0] += 1
    v[1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in c.values())
print("%d calls, %.2f ms of event time" % (len(tr), tot / 1e3))
for (name, scal), (cnt, us) in sorted(c.items(), key=lambda kv: -kv[1][1]):
    if us < floor:
        continue
    print("%8.1f us  x%-3d avg %7.1f  %s %s" % (us, cnt, us / cnt, name.replace("trid_", ""), " ".join(str(a) for a in scal)))
