#!/bin/bash
mkdir -p gpurun_out/r05_soak
python tools/soak.py 2500 --replay 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_soak/replay.txt | awk 'NR<=4 || NR%10==0'
