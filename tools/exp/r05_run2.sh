#!/bin/bash
# round 5, run 2: eval-mode encoder on the P16 kernels - kernel tests, model tests, timing
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_run2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "eval or tile96" > $OUT/tests_k.txt 2>&1; tail -15 $OUT/tests_k.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "visual_encoder or eval" > $OUT/tests_m.txt 2>&1; tail -15 $OUT/tests_m.txt
timeout 600 python tools/eval_time.py rn50 128 32 > $OUT/eval_time.txt 2>&1; cat $OUT/eval_time.txt
