"""GPU box: where the HOST time of an eager train step goes (cProfile over 5 steps; the GPU side runs unobserved)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from textreid_amd.caption import CaptionBatch
from textreid_amd.config import moco_cfg
from textreid_amd.model import build_model
from textreid_amd.solver import make_optimizer
dev = torch.device("cuda"); torch.manual_seed(0)
cfg = moco_cfg("m_resnet50", K=8192)
model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(dev); model.train()
opt = make_optimizer(cfg, model)
B = 128
batches = [bench.synth_batch(B, s, dev, 1234) for s in range(2)]
def step(i):
    images, tokens, lengths, ids = batches[i % 2]
    ld = model(images, CaptionBatch(tokens, lengths, ids % 11003, max_len=64))
    opt.zero_grad(); sum(ld.values()).backward(); opt.step()
for i in range(4): step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(5): step(i)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue %.1f ms per step (un-profiled)" % (th / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
