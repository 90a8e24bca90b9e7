#!/bin/bash
# r06k: BatchNorm passes with the (row, channel quad) carried along and several elements in flight, against the build before
# (textreid_amd/libtextreid_hip_base.so): isolated passes, the BatchNorm tests, bench.py on both libraries (same box)
O=gpurun_out/r06k; mkdir -p $O
TRID_LIB_PATH=textreid_amd/libtextreid_hip_base.so timeout 300 python tools/bn_p16_bench.py > $O/bn_base.txt 2>&1
timeout 300 python tools/bn_p16_bench.py > $O/bn_new.txt 2>&1
grep -h "total" $O/bn_base.txt $O/bn_new.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -m gpu -k "bn or batchnorm or BatchNorm or bottleneck" > $O/bn_tests.txt 2>&1; tail -2 $O/bn_tests.txt
TRID_LIB_PATH=textreid_amd/libtextreid_hip_base.so timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err
TRID_LIB_PATH=textreid_amd/libtextreid_hip_base.so timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_base2.json 2> $O/bench_base2.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_new2.json 2> $O/bench_new2.err
for f in base new base2 new2; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d.get('replay_equals_eager_b128'))"; done
