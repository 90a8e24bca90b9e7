"""Choose the seeds of tests/test_model_gpu.py's full-size one-step cases: the first seed whose query
image encoder has a smallest |ReLU input| >= oracle.fill.RELU_MIN on the fp32 CPU oracle (see the
`margin` fill style in oracle/fill.py).  CPU only.

  python tools/pick_fullstep_seed.py rn50 16 64 21      # spec, B, K, first seed to try
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.fill as OF  # noqa: E402
import oracle.head as OH  # noqa: E402
import oracle.visual as OV  # noqa: E402
from oracle.cases import full_step_case, relu_floor  # noqa: E402

spec = {"rn50": OV.RN50, "rn101": OV.RN101}[sys.argv[1]]
B, K, start = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
for seed in range(start, start + 100):
    st, table, images, tokens, lengths, ids = full_step_case(spec, B, K, 3000, seed)
    taps = {}
    with torch.no_grad():
        OV.visual_forward(OH._Sub(st, "v_encoder_q"), images, spec, True, taps)
    print("seed %d: smallest |ReLU input| %.2e" % (seed, taps["relu_min"]), flush=True)
    if taps["relu_min"] >= relu_floor(B):
        break
