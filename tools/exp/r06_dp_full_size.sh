#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
export PYTHONFAULTHANDLER=1 TRID_REPLAY_DEBUG=1
timeout 2500 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "four_ranks or eight_ranks" --durations=5 > $O/dp_full_tests.txt 2>&1; tail -12 $O/dp_full_tests.txt | cut -c1-300
