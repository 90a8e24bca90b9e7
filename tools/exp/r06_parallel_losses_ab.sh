#!/bin/bash
# r06o: the three losses on their own streams (TRID_PARALLEL_LOSSES): head / capture tests, then three A/B rounds of the step on one box
O=gpurun_out/r06o; mkdir -p $O
timeout 1500 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "not b128" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do
TRID_PARALLEL_LOSSES=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_off$i.json 2> $O/bench_off$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_on$i.json 2> $O/bench_on$i.err
done
for f in off1 on1 off2 on2 off3 on3; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); lp=d['config']['launch_probe']; print('$f', round(d['ms_per_step'],3), lp['chosen'], round(lp['stream_replay_ms_per_step'],3), round(lp['eager_ms_per_step'],3), lp['stream_replay_plan']['lanes'], d.get('replay_equals_eager_b128'))"; done
