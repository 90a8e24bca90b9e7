#!/bin/bash
# r06m: bn3's backward sums from the GEMM that produces the block-output gradient (TRID_BN3_FUSE): tests, then same-box A/B of the step
O=gpurun_out/r06m; mkdir -p $O
timeout 1200 python -m pytest tests/test_blocks_gpu.py tests/test_match_state_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "switches or bottleneck or captured_train_step_equals or config1_b128_k8192 or visual_encoder_full_size or full_size_step" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2; do
TRID_BN3_FUSE=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_off$i.json 2> $O/bench_off$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_on$i.json 2> $O/bench_on$i.err
done
for f in off1 on1 off2 on2; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d.get('replay_equals_eager_b128'), d['config']['launch_probe']['stream_replay_plan']['kernels'])"; done
