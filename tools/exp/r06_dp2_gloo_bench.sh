#!/bin/bash
# r06v: bench.py --gpus 2 with the gloo transport, both ranks on the one GPU: the data-parallel step's segmented replay vs its eager form under bench.py itself
O=gpurun_out/r06v; mkdir -p $O
TRID_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp2_segmented.json 2> $O/bench_dp2_segmented.err
TRID_DIST_BACKEND=gloo TRID_DP_CAPTURE=0 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp2_eager.json 2> $O/bench_dp2_eager.err
for f in dp2_segmented dp2_eager; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['n_gpus'], d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'], d['config'].get('host_work_ms_per_step_no_backpressure'), d['config']['step_launch'][:70], d.get('data_parallel'))"; done; tail -3 $O/bench_dp2_segmented.err
