#!/bin/bash
# r06t: the 192-row one-workgroup tile chosen by the library for the half-filling forward shapes (TRID_P16_TILE192): tests, then the step A/B
O=gpurun_out/r06t; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_model_gpu.py tests/test_match_state_gpu.py -x -q -m gpu -k "tile96 or bottleneck or config1_b128 or visual_encoder_full_size or b128_replay or captured_train_step_equals or eval" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do
TRID_P16_TILE192=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_off$i.json 2> $O/bench_off$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_on$i.json 2> $O/bench_on$i.err
done
for f in off1 on1 off2 on2 off3 on3; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); lp=d['config']['launch_probe']; print('$f', round(d['ms_per_step'],3), lp['chosen'], round(lp['stream_replay_ms_per_step'],3), d.get('replay_equals_eager_b128'))"; done
