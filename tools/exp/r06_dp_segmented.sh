#!/bin/bash
# r06f: the data-parallel step on the segmented replay (tests/test_dp_gpu.py), bucketed recordings, the host path of a one-rank nccl
# bench (TRID_DP_FORCE=1) segmented vs eager, the 1x1 shapes after the stats-epilogue change.  Every leg under its own timeout.
O=gpurun_out/r06f; mkdir -p $O
export TRID_REPLAY_DEBUG=1
timeout 1700 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "not eight_ranks" > $O/dp_tests.txt 2>&1; tail -5 $O/dp_tests.txt
unset TRID_REPLAY_DEBUG
timeout 600 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "bucketed or loose_caption or do_train_captured or b128_replay" > $O/bucket_tests.txt 2>&1; tail -3 $O/bucket_tests.txt
TRID_DP_FORCE=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_segmented.json 2> $O/bench_dp1_segmented.log
TRID_DP_FORCE=1 TRID_DP_CAPTURE=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_eager.json 2> $O/bench_dp1_eager.log
for f in dp1_segmented dp1_eager; do python - $O/bench_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", d["ms_per_step"], "host ms/step", d["config"].get("host_enqueue_ms_per_step"), d["config"].get("step_launch","")[:120], d.get("data_parallel"))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
done
timeout 300 python tools/kloop_bench.py 3 12 2>&1 | grep -A14 "3x3 total" > $O/kloop_1x1.txt; tail -3 $O/kloop_1x1.txt
