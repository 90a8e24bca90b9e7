#!/bin/bash
# round 5, run 1: 96-row tile variant - parity test, isolated shapes, same-box A/B of the step (TRID_P16_TILE96=0 vs 1)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05_run1; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "tile96 or stream or fused or halo" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
python tools/exp/tile96_bench.py > $OUT/tile96_bench.txt 2>&1; cat $OUT/tile96_bench.txt
for i in 1 2; do
  for v in 0 1; do
    TRID_P16_TILE96=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_P16_TILE96=$v ms_per_step %.2f' % d['ms_per_step'])" | tee -a $OUT/ab.txt
  done
done
