#!/bin/bash
# round 5, run 32: data-gradient filter forms packed during the forward instead of at the head of backward (TRID_EARLY_WPT)
mkdir -p gpurun_out/r05_run32
O=gpurun_out/r05_run32
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full or config1 or step or determin" > $O/t2.txt 2>&1; tail -2 $O/t2.txt
for i in 1 2 3; do for v in 1 0; do
  TRID_BENCH_LAUNCH=streams TRID_EARLY_WPT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_EARLY_WPT=$v ms_per_step %.2f' % d['ms_per_step'])" | tee -a $O/ab.txt
done; done
