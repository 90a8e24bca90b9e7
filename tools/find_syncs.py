#!/usr/bin/env python3
"""Host<->device synchronisation points of one training step (torch's sync debug mode, warnings with their call
sites) and the host-side enqueue time of a step against its device time.  GPU box:  python tools/find_syncs.py"""
import os, sys, time, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from textreid_amd.caption import CaptionBatch
from textreid_amd.config import moco_cfg
from textreid_amd.model import build_model
from textreid_amd.solver import make_optimizer

dev = torch.device("cuda")
torch.manual_seed(0)
cfg = moco_cfg("m_resnet50", K=8192)
model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(dev)
model.train()
opt = make_optimizer(cfg, model)
B = 128
batches = [bench.synth_batch(B, s, dev, 1234) for s in range(4)]


def step(i):
    images, tokens, lengths, ids = batches[i % 4]
    cb = CaptionBatch(tokens, lengths, ids % 11003, max_len=64)
    losses = sum(model(images, cb).values())
    opt.zero_grad()
    losses.backward()
    opt.step()
    return losses


for i in range(3):
    step(i)
torch.cuda.synchronize()
sites = collections.Counter()
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        st = [f for f in traceback.extract_stack() if "/textreid_amd/" in f.filename or f.filename.endswith("find_syncs.py")]
        sites[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-4:]))] += 1
    else:
        orig(message, category, filename, lineno, file, line)


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step(3)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising calls in one step: %d" % sum(sites.values()))
for k, v in sites.most_common():
    print("  %3d x %s" % (v, k))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(4 + rep)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.1f ms, device done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
