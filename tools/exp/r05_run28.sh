#!/bin/bash
# round 5, run 28: a residual block's weight gradients behind ONE event record on the main stream (TRID_WGRAD_BATCH): tests, step A/B
mkdir -p gpurun_out/r05_run28
O=gpurun_out/r05_run28
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full or config1 or block or step or determin" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "captur or determin or do_train" > $O/t3.txt 2>&1; tail -3 $O/t3.txt
for i in 1 2 3; do for v in 1 0; do
  TRID_WGRAD_BATCH=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_WGRAD_BATCH=$v ms_per_step %.2f (%s) events %s' % (d['ms_per_step'], d['config']['launch_probe']['chosen'], d['config']['launch_probe']['stream_replay_plan']['events']))" | tee -a $O/ab.txt
done; done
