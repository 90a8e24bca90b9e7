#!/usr/bin/env python3
"""A few configs[3]-shape steps (RN101, K=65536, B=128) for rocprofv3 --kernel-trace --stats.  usage: c3_prof.py <prec>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from textreid_amd import ops
from textreid_amd.caption import CaptionBatch
from textreid_amd.config import moco_cfg
from textreid_amd.model import build_model
from textreid_amd.solver import make_optimizer
dev = torch.device("cuda"); B = 128
ops.CONV_PRECISION = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
cfg = moco_cfg("m_resnet101", K=65536)
model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(dev).train()
opt = make_optimizer(cfg, model)
batches = [bench.synth_batch(B, s, dev, 4321) for s in range(2)]
for i in range(10):
    images, tokens, lengths, ids = batches[i % 2]
    ld = model(images, CaptionBatch(tokens, lengths, (ids + i * (B // 4)) % 11003, max_len=64)); opt.zero_grad(); sum(ld.values()).backward(); opt.step()
torch.cuda.synchronize()
