#!/bin/bash
# round 5, run 29: how many residual blocks share one weight-gradient event (TRID_WGRAD_BATCH = 1, 2, 3, 4)
mkdir -p gpurun_out/r05_run29
O=gpurun_out/r05_run29
for i in 1 2; do for v in 1 2 3 4; do
  TRID_BENCH_LAUNCH=streams TRID_WGRAD_BATCH=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_WGRAD_BATCH=$v ms_per_step %.2f events %s' % (d['ms_per_step'], d['config']['launch_probe']['stream_replay_plan']['events']))" | tee -a $O/ab.txt
done; done
