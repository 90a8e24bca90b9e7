#!/bin/bash
# round 5, run 8: candidate staging in the retrieval filter kernel - tests, timing, bench
mkdir -p gpurun_out/r05_run8
python -m pytest tests/test_match_state_gpu.py -x -q -m gpu > gpurun_out/r05_run8/tests.txt 2>&1
tail -3 gpurun_out/r05_run8/tests.txt
python tools/retrieval_time.py 1000000 > gpurun_out/r05_run8/retrieval_time.txt 2>&1
cat gpurun_out/r05_run8/retrieval_time.txt
