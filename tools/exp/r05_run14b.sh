#!/bin/bash
python tools/exp/replay_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -5
