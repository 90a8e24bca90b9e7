#!/bin/bash
# r06l: the division-free BatchNorm loops (no ILP) against the build before, in the step; BatchNorm tests
O=gpurun_out/r06l; mkdir -p $O
timeout 300 python tools/bn_p16_bench.py > $O/bn_new.txt 2>&1; grep -h "total" $O/bn_new.txt
for i in 1 2; do
TRID_LIB_PATH=textreid_amd/libtextreid_hip_base.so timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_base$i.json 2> $O/bench_base$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_new$i.json 2> $O/bench_new$i.err
done
for f in base1 new1 base2 new2; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d.get('replay_equals_eager_b128'))"; done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -m gpu -k "bn or batchnorm or BatchNorm or bottleneck" > $O/bn_tests.txt 2>&1; tail -2 $O/bn_tests.txt
