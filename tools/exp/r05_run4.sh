#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_run4; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "eval or halo or stem_conv1 or pooled" > $OUT/tests_k.txt 2>&1; tail -5 $OUT/tests_k.txt; timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "visual_encoder or eval" > $OUT/tests_m.txt 2>&1; tail -5 $OUT/tests_m.txt
timeout 600 python tools/eval_time.py rn50 128 --only-p16 > $OUT/eval_time.txt 2>&1; cat $OUT/eval_time.txt
for b in 128; do timeout 300 python tools/eval_calls.py rn50 $b > $OUT/eval_calls_$b.txt 2>&1; done
head -60 $OUT/eval_calls_128.txt
