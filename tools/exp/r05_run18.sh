#!/bin/bash
mkdir -p gpurun_out/r05_run18
python tools/exp/bn_bw_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_run18/bn_bw.txt
