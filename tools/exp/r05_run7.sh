#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_run7; mkdir -p $OUT
free -g | head -2; nproc
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -s -k "unselected or he_style or losses" > $OUT/tests_m.txt 2>&1; grep -E "he-style|B=128 he|passed|failed|Error" $OUT/tests_m.txt | tail -8
timeout 1200 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -s -k "do_train or two_ranks_match" > $OUT/tests_dp.txt 2>&1; tail -6 $OUT/tests_dp.txt
timeout 600 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "failed_capture" > $OUT/tests_cap.txt 2>&1; tail -3 $OUT/tests_cap.txt
