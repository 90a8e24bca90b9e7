#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "gemm_p16_stream" > $OUT/t_stream.log 2>&1; echo "stream rc=$?"
tail -25 $OUT/t_stream.log | cut -c1-300
timeout 600 python tools/stream_bench.py > $OUT/stream_bench.txt 2>&1; echo "bench rc=$?"
cat $OUT/stream_bench.txt
