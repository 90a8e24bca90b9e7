"""What the recorded train step holds (nodes by type, lanes, events) and whether its stream replay runs - csrc/step_replay.hip."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from textreid_amd.caption import CaptionBatch
from textreid_amd.config import moco_cfg
from textreid_amd.engine.graph import CapturedTrainStep
from textreid_amd.model import build_model
from textreid_amd.solver import make_optimizer

gpu = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = moco_cfg("m_resnet50", K=64 if B < 64 else 8192)
table = torch.randn(3000, 512, generator=torch.Generator().manual_seed(1)) * 0.02
torch.manual_seed(0)
model = build_model(cfg, vocab_dict=table).to(gpu).train()
opt = make_optimizer(cfg, model)
runner = CapturedTrainStep(model, opt, warmup=2, caption_bound=64)
for i in range(5):
    images, tokens, lengths, ids = bench.synth_batch(B, i, gpu, 5, vocab=3000)
    cb = CaptionBatch(tokens, lengths, ids % 11003, max_len=64)
    try:
        ld = runner(images, cb)
        torch.cuda.synchronize()
        print(i, "ok", {k: float(v) for k, v in ld.items()}, runner.replay_info)
    except RuntimeError as e:
        print(i, "FAILED", str(e)[:400], runner.replay_info)
        break

import hashlib
torch.cuda.synchronize()
h = hashlib.sha256()
for n_, p_ in model.named_parameters():
    h.update(p_.detach().cpu().numpy().tobytes())
print("parameter digest after the steps:", h.hexdigest()[:16])
# host time of one replay call with the GPU idle at the start (is the call slow, or does it wait for queue space?)
import time
if runner.replayer is not None:
    from textreid_amd import ops
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.call("trid_step_replay_run", runner.replayer, ops.stream())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("replay call: host %.2f ms, until the GPU is done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
