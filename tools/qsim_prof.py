"""Profile helper: the queue-similarity / InfoNCE block alone (bench.queue_similarity_bench) at K=8192 and 65536."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for K in (8192, 65536):
    print(bench.queue_similarity_bench(torch.device("cuda"), K=K, reps=50))
