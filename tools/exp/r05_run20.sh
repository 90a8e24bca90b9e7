#!/bin/bash
mkdir -p gpurun_out/r05_run20
python tools/exp/eval_two_streams.py 256 512 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_run20/two_streams.txt
