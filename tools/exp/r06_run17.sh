#!/bin/bash
# r06s: full GPU suite at HEAD; layer1's conv3 data gradient on the tile kernel with the BatchNorm-backward sums (default) vs on the streaming kernel + reduce pass
O=gpurun_out/r06s; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do
TRID_BNB_FUSE_STREAM_SHAPES=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_off$i.json 2> $O/bench_off$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_on$i.json 2> $O/bench_on$i.err
done
for f in off1 on1 off2 on2 off3 on3; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); lp=d['config']['launch_probe']; print('$f', round(d['ms_per_step'],3), lp['chosen'], round(lp['stream_replay_ms_per_step'],3), lp['stream_replay_plan']['kernels'])"; done
