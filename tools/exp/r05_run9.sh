#!/bin/bash
# round 5, run 9: the retrieval filter kernel - DMA pieces spread over the k loop (flag 1), upper waves scanning one step late (flag 2)
mkdir -p gpurun_out/r05_run9
: > gpurun_out/r05_run9/out.txt
python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "topk or retrieval or sim" 2>&1 | tail -1 >> gpurun_out/r05_run9/out.txt
for fl in 0 1 2 3; do
  echo "TRID_TOPK_FLAGS=$fl" >> gpurun_out/r05_run9/out.txt
  TRID_TOPK_FLAGS=$fl TRID_RETR_ONLY_P16=1 python tools/retrieval_time.py 1000000 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_run9/out.txt
done
TRID_TOPK_FLAGS=3 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "topk or retrieval or sim" 2>&1 | tail -1 >> gpurun_out/r05_run9/out.txt
cat gpurun_out/r05_run9/out.txt
