#!/bin/bash
# r06g: where do the data-parallel runs die?  (faulthandler stacks)
O=gpurun_out/r06g; mkdir -p $O
export PYTHONFAULTHANDLER=1
TRID_DP_FORCE=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_segmented.json 2> $O/bench_dp1_segmented.log; echo "rc=$?" >> $O/bench_dp1_segmented.log
TRID_DP_FORCE=1 TRID_DP_CAPTURE=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_eager.json 2> $O/bench_dp1_eager.log; echo "rc=$?" >> $O/bench_dp1_eager.log
timeout 600 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "replays_in_segments" > $O/dp_tests.txt 2>&1
tail -30 $O/bench_dp1_segmented.log; tail -30 $O/bench_dp1_eager.log
