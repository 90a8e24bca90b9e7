#!/bin/bash
# round 5, run 35: quad-major layout of the BatchNorm-backward partials (the fold reads whole lines): tests, step A/B vs the previous library
mkdir -p gpurun_out/r05_run35
O=gpurun_out/r05_run35
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bn or two_layers or backward_sums" > $O/t1.txt 2>&1; tail -2 $O/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full or config1 or step or block" > $O/t2.txt 2>&1; tail -2 $O/t2.txt
for i in 1 2 3; do
  TRID_BENCH_LAUNCH=streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ms_per_step %.2f' % d['ms_per_step'])" | tee -a $O/ab.txt
  TRID_LIB_PATH=$PWD/_ab/lib_prev.so TRID_BENCH_LAUNCH=streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old ms_per_step %.2f' % d['ms_per_step'])" | tee -a $O/ab.txt
done
