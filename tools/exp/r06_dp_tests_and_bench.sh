#!/bin/bash
# r06h: the data-parallel tests on the segmented replay incl. the 4-rank full-size case; one-rank nccl bench segmented vs eager
O=gpurun_out/r06h; mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 2000 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "not eight_ranks" --durations=10 > $O/dp_tests.txt 2>&1; tail -16 $O/dp_tests.txt
TRID_DP_FORCE=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_segmented.json 2> $O/bench_dp1_segmented.log
TRID_DP_FORCE=1 TRID_DP_CAPTURE=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_eager.json 2> $O/bench_dp1_eager.log
for f in dp1_segmented dp1_eager; do python - $O/bench_$f.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "ms/step", d["ms_per_step"], "host ms/step", d["config"].get("host_enqueue_ms_per_step"), d["config"].get("step_launch","")[:100], d.get("data_parallel"))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
done
