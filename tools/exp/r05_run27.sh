#!/bin/bash
# round 5, run 27: the key encoder held a fixed number of kernels behind the query encoder in the stream plan (TRID_REPLAY_PP)
mkdir -p gpurun_out/r05_run27
O=gpurun_out/r05_run27/ab.txt; : > $O
for pp in none 0,2,0,2 0,2,1,3 0,2,2,4 0,2,3,6 none 0,2,1,4; do
  if [ $pp = none ]; then unset TRID_REPLAY_PP; else export TRID_REPLAY_PP=$pp; fi
  TRID_BENCH_LAUNCH=streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PP=$pp ms_per_step %.2f  plan %s' % (d['ms_per_step'], d['config']['launch_probe']['stream_replay_plan']))" | tee -a $O
done
