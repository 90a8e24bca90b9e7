#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "fused_queue_infonce" > $OUT/t_q.log 2>&1; echo "queue rc=$?"; tail -3 $OUT/t_q.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "moco_head or losses or head_step" > $OUT/t_h.log 2>&1; echo "head rc=$?"; tail -3 $OUT/t_h.log
timeout 900 python -m pytest tests/test_match_state_gpu.py -q -x -k "captured or deterministic" > $OUT/t_c.log 2>&1; echo "capt rc=$?"; tail -3 $OUT/t_c.log
timeout 600 python tools/qsim_tune.py > $OUT/qsim_tune.txt 2>&1; cat $OUT/qsim_tune.txt
