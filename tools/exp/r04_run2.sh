#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "halo or stem_conv1" > $OUT/t_halo.log 2>&1; echo "halo rc=$?"
tail -25 $OUT/t_halo.log
timeout 600 python tools/stem_bench.py > $OUT/stem_bench.txt 2>&1; echo "bench rc=$?"
cat $OUT/stem_bench.txt
