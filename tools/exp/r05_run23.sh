#!/bin/bash
# round 5, run 23: paired whole-line P16 LOADS of the identity branch / pooled inputs in the elementwise passes: tests, step A/B
mkdir -p gpurun_out/r05_run23
O=gpurun_out/r05_run23
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bn or pool or block or p16" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
  TRID_LIB_PATH=$PWD/_ab/lib_prev.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
done
