#!/bin/bash
# round 5, run 38: scope of the release that the stream plan's event records carry (TRID_REPLAY_EVFLAGS)
mkdir -p gpurun_out/r05_run38
O=gpurun_out/r05_run38/ab.txt; : > $O
for i in 1 2; do for v in 0 1 2 3; do
  TRID_BENCH_LAUNCH=streams TRID_REPLAY_EVFLAGS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('EVFLAGS=$v ms_per_step %.2f final_loss %.6f' % (d['ms_per_step'], d['config']['final_loss']))" | tee -a $O
done; done
