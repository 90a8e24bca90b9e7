#!/bin/bash
# round 5, run 15: stream replay of the step with stream priorities (lane 0 = the recording stream's chain)
mkdir -p gpurun_out/r05_run15
O=gpurun_out/r05_run15/ab.txt; : > $O
for i in 1 2; do for pr in 0 1 2; do
TRID_REPLAY_PRIO=$pr TRID_BENCH_LAUNCH=streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prio=$pr ms_per_step %.2f  host %.1f' % (d['ms_per_step'], d['config']['host_enqueue_ms_per_step']))" | tee -a $O
done; done
