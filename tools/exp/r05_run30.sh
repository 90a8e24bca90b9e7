#!/bin/bash
# round 5, run 30: the key encoder started k kernels after the query encoder (ONE extra edge in the stream plan)
mkdir -p gpurun_out/r05_run30
O=gpurun_out/r05_run30/ab.txt; : > $O
for off in none 0,2,8 0,2,20 0,2,40 0,2,80 none 0,2,120; do
  if [ $off = none ]; then unset TRID_REPLAY_OFFSET; else export TRID_REPLAY_OFFSET=$off; fi
  TRID_BENCH_LAUNCH=streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('OFFSET=$off ms_per_step %.2f  events %s' % (d['ms_per_step'], (d['config']['launch_probe']['stream_replay_plan'] or {}).get('events')))" | tee -a $O
done
