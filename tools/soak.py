#!/usr/bin/env python3
"""Soak run: N training steps of configs[1] on synthetic data; prints loss, step time and allocator statistics every 50
steps (memory growth, non-finite losses and step-time drift show up here, not in a 10-step bench).
GPU box:  python tools/soak.py [steps] [--replay | --buckets]   (--replay: the recorded step re-issued as stream launches, engine/graph.py;
--buckets: ragged captions of 8-64 tokens through engine.graph.BucketedTrainStep, one recording per caption bucket, as do_train runs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from textreid_amd.caption import CaptionBatch
from textreid_amd.config import moco_cfg
from textreid_amd.model import build_model
from textreid_amd.solver import make_optimizer

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
REPLAY = "--replay" in sys.argv
BUCKETS = "--buckets" in sys.argv
dev = torch.device("cuda"); torch.manual_seed(0)
cfg = moco_cfg("m_resnet50", K=8192)
model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(dev); model.train()
opt = make_optimizer(cfg, model)
B = 128
batches = [bench.synth_batch(B, s, dev, 1234) for s in range(8)]
runner = None
if REPLAY:
    from textreid_amd.engine.graph import CapturedTrainStep
    runner = CapturedTrainStep(model, opt, warmup=2, caption_bound=64)
if BUCKETS:
    from textreid_amd.engine.graph import BucketedTrainStep
    runner = BucketedTrainStep(model, opt, warmup=2)
    gen = torch.Generator().manual_seed(11)
    ragged = []
    for s, (im, tk, ln, ids) in enumerate(batches):  # every batch its own longest caption: 20 ... 64 tokens
        top = (20, 31, 40, 47, 56, 64, 28, 44)[s]
        ln2 = torch.randint(8, top + 1, (B,), generator=gen); ln2[s] = top
        tk2 = tk.cpu().clone()
        for r, n in enumerate(ln2.tolist()): tk2[r, n:] = 0
        ragged.append((im, tk2.to(dev), ln2.to(dev), ids, top))
t0 = time.perf_counter(); last = None
for i in range(N):
    if BUCKETS:
        images, tokens, lengths, ids, top = ragged[i % 8]
        cb = CaptionBatch(tokens, lengths, (ids + (i // 8) * 8 * (B // 4)) % 11003, max_len=top)
    else:
        images, tokens, lengths, ids = batches[i % 8]
        cb = CaptionBatch(tokens, lengths, (ids + (i // 8) * 8 * (B // 4)) % 11003, max_len=64)
    if runner is not None:
        ld = runner(images, cb)
    else:
        ld = model(images, cb)
        loss = sum(ld.values()); opt.zero_grad(); loss.backward(); opt.step()
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e3; t0 = time.perf_counter()
        vals = {k: float(v) for k, v in ld.items()}
        ok = all(v == v and abs(v) < 1e6 for v in vals.values())
        print("step %4d  %.2f ms/step  alloc %.3f GB in %d blocks  reserved %.2f GB  losses %s%s" % (
            i + 1, dt, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_stats()["allocation.all.current"],
            torch.cuda.memory_reserved() / 2**30,
            {k: round(v, 4) for k, v in vals.items()}, "" if ok else "  NON-FINITE") + ("  recordings %s" % runner.recorded if BUCKETS else ""), flush=True)
