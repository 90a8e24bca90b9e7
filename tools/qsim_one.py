"""The fused queue-similarity / InfoNCE block at ONE queue length (for counter passes):
python tools/qsim_one.py K [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = bench.queue_similarity_bench(torch.device("cuda"), K=K, reps=reps)
print({k: v for k, v in r.items() if k != "note"})
