#!/bin/bash
# round 5, run 34: the attention pool's projection-weight gradients on the weight-gradient stream: tests, step (new build vs TRID_SERIAL_ATTN_WGRAD=1)
mkdir -p gpurun_out/r05_run34
O=gpurun_out/r05_run34
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full or config1 or step or determin or attn" > $O/t2.txt 2>&1; tail -2 $O/t2.txt
python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "captur or determin or do_train" > $O/t3.txt 2>&1; tail -2 $O/t3.txt
for i in 1 2 3; do for v in 0 1; do
  TRID_BENCH_LAUNCH=streams TRID_SERIAL_ATTN_WGRAD=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_SERIAL_ATTN_WGRAD=$v ms_per_step %.2f' % d['ms_per_step'])" | tee -a $O/ab.txt
done; done
