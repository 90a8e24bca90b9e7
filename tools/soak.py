import sys, time, torch
sys.path.insert(0, '.')
import bench
from textreid_amd.caption import CaptionBatch
from textreid_amd.config import moco_cfg
from textreid_amd.model import build_model
from textreid_amd.solver import make_optimizer
dev = torch.device("cuda")
torch.manual_seed(0)
cfg = moco_cfg("m_resnet50", K=8192)
model = build_model(cfg, vocab_dict=torch.randn(49408, 512) * 0.02).to(dev).train()
opt = make_optimizer(cfg, model)
B = 128
batches = [bench.synth_batch(B, s, dev, 1234) for s in range(8)]
t0 = time.time(); hist = []
for i in range(400):
    im, tk, ln, ids = batches[i % 8]
    ld = model(im, CaptionBatch(tk, ln, (ids + (i // 8) * 8 * 32) % 11003, max_len=64))
    loss = sum(ld.values())
    opt.zero_grad(); loss.backward(); opt.step()
    if i % 50 == 0 or i == 399:
        v = {k: float(x) for k, x in ld.items()}
        assert all(x == x and abs(x) < 1e6 for x in v.values()), v
        hist.append((i, round(sum(v.values()), 3), round(torch.cuda.max_memory_allocated() / 2**30, 2)))
torch.cuda.synchronize()
print("400 steps in %.1fs" % (time.time() - t0)); print(hist)
