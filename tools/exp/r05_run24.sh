#!/bin/bash
mkdir -p gpurun_out/r05_run24
python tools/exp/lo_bits_power.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_run24/lo_bits.txt
