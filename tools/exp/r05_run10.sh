#!/bin/bash
# round 5, run 10: segmented retrieval filter (+ staging, late scan): tests, timing A/B
mkdir -p gpurun_out/r05_run10
O=gpurun_out/r05_run10/out.txt; : > $O
TRID_TOPK_FLAGS=3 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $O
for seg in 0 1; do for fl in 0 3; do
  echo "TRID_TOPK_SEGMENTS=$seg TRID_TOPK_FLAGS=$fl" >> $O
  TRID_TOPK_SEGMENTS=$seg TRID_TOPK_FLAGS=$fl TRID_RETR_ONLY_P16=1 python tools/retrieval_time.py 1000000 2>&1 | grep -v amdgpu.ids >> $O
done; done
cat $O
