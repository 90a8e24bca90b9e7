"""Profile helper (run under rocprofv3 --kernel-trace --stats): the fused queue-similarity / InfoNCE
block alone at K = 8192 and K = 65536, 30 calls each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for K in (8192, 65536):
    print(bench.queue_similarity_bench(torch.device("cuda"), K=K, reps=30))
