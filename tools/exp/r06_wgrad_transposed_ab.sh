#!/bin/bash
# r06q: weight-gradient GEMM with the transposed accumulator (16-byte slab stores) against the build before: digests, timings, tests, step A/B
O=gpurun_out/r06q; mkdir -p $O
TRID_LIB_PATH=textreid_amd/libtextreid_hip_base.so timeout 300 python tools/kloop_bench.py --wgrad > $O/wgrad_base.txt 2>&1
timeout 300 python tools/kloop_bench.py --wgrad > $O/wgrad_new.txt 2>&1
paste -d'|' <(grep -v amdgpu $O/wgrad_base.txt | cut -c1-70) <(grep -v amdgpu $O/wgrad_new.txt | cut -c30-70)
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q -m gpu -k "wgrad or bottleneck or bn3" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for i in 1 2 3; do
TRID_LIB_PATH=textreid_amd/libtextreid_hip_base.so timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_base$i.json 2> $O/bench_base$i.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_new$i.json 2> $O/bench_new$i.err
done
for f in base1 new1 base2 new2 base3 new3; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); lp=d['config']['launch_probe']; print('$f', round(d['ms_per_step'],3), lp['chosen'], round(lp['stream_replay_ms_per_step'],3), d.get('replay_equals_eager_b128'))"; done
