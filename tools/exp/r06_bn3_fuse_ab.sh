#!/bin/bash
# r06n: bn3 sums from the producing GEMM - off / layer4 only (planes >= 512) / layer3 + layer4 (planes >= 256), three rounds on one box
O=gpurun_out/r06n; mkdir -p $O
for i in 1 2 3; do
TRID_BN3_FUSE=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_off$i.json 2> $O/bench_off$i.err
TRID_BN3_FUSE_MIN_PLANES=512 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_p512_$i.json 2> $O/bench_p512_$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_p256_$i.json 2> $O/bench_p256_$i.err
done
for f in off1 p512_1 p256_1 off2 p512_2 p256_2 off3 p512_3 p256_3; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', round(d['ms_per_step'],3), d['config']['launch_probe']['chosen'], round(d['config']['launch_probe']['stream_replay_ms_per_step'],3), d['config']['launch_probe']['stream_replay_plan']['kernels'])"; done
