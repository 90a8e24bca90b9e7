#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_run5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "similarity_topk or topk_full or config4" > $OUT/tests.txt 2>&1; tail -8 $OUT/tests.txt
timeout 600 python tools/retrieval_time.py > $OUT/retrieval_time.txt 2>&1; cat $OUT/retrieval_time.txt
