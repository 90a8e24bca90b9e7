#!/bin/bash
mkdir -p gpurun_out/r05_run10
python -m pytest tests/test_match_state_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05_run10/fail.txt
cat gpurun_out/r05_run10/fail.txt
