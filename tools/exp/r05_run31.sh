#!/bin/bash
# round 5, run 31: weight gradients of alternate blocks on two side streams (TRID_WGRAD_STREAMS)
mkdir -p gpurun_out/r05_run31
O=gpurun_out/r05_run31/ab.txt; : > $O
for i in 1 2; do for v in 1 2 3; do
  TRID_BENCH_LAUNCH=streams TRID_WGRAD_STREAMS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_WGRAD_STREAMS=$v ms_per_step %.2f  plan %s' % (d['ms_per_step'], d['config']['launch_probe']['stream_replay_plan']))" | tee -a $O
done; done
