"""GPU box: can RCCL collectives be recorded into a hipGraph on this stack (torch 2.10 + ROCm 7.2)?  One-rank `nccl` group:
all_gather_into_tensor + async all_reduce (work.wait()) inside torch.cuda.graph, replayed with new inputs."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group("nccl", init_method="env://")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
x = torch.arange(8, dtype=torch.float32, device=dev)
out = torch.empty(8, device=dev)
g2 = torch.ones(1 << 20, device=dev)
# warm-up (communicator creation is not capturable)
dist.all_gather_into_tensor(out, x)
dist.all_reduce(g2)
torch.cuda.synchronize()
side = torch.cuda.Stream()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2
        dist.all_gather_into_tensor(out, y)
        z = out + 1
        w = dist.all_reduce(g2, async_op=True)
        q = z * 3
        w.wait()
        r = g2 * 0.5 + q.sum()
    torch.cuda.synchronize()
    for k in range(3):
        x.copy_(torch.arange(8, device=dev) + 10.0 * k)
        g2.fill_(float(k + 1))
        g.replay()
        torch.cuda.synchronize()
        want_q = ((torch.arange(8) + 10.0 * k) * 2 + 1) * 3
        assert torch.equal(q.cpu(), want_q), (q, want_q)
        assert torch.allclose(r.cpu(), torch.full((1 << 20,), (k + 1) * 0.5) + want_q.sum()), r[:4]
    print("RCCL_CAPTURE_OK: all_gather_into_tensor + async all_reduce recorded and replayed 3x")
except Exception as e:  # noqa: BLE001
    print("RCCL_CAPTURE_FAILED:", type(e).__name__, str(e)[:2000])
    sys.exit(0)
dist.destroy_process_group()
