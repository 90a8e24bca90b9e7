#!/bin/bash
# round 5, run 13: same-box A/B of the train step, driver flags (--steps 20 --warmup 5): current library vs one whose streaming
# kernels are the ones before the compiler-visible wait / retrieval rework
mkdir -p gpurun_out/r05_run13
O=gpurun_out/r05_run13/ab.txt; : > $O
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export TRID_LIB_PATH=$PWD/_ab/lib_oldstream.so; else unset TRID_LIB_PATH; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v ms_per_step %.2f  host %.1f' % (d['ms_per_step'], d['config']['host_enqueue_ms_per_step']))" | tee -a $O
  done
done
