#!/bin/bash
# round 5, run 22: whole-line P16 stores in the streaming kernel's fused epilogue and the tile kernel's eval epilogue: tests, eval rate, step A/B
mkdir -p gpurun_out/r05_run22
O=gpurun_out/r05_run22
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "eval or fused or stream or conv1x1 or block" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "eval or golden or full" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
for i in 1 2; do
  python tools/eval_time.py rn50 128 512 --only-p16 2>&1 | grep "P16 eval" | sed 's/^/new /' | tee -a $O/eval.txt
  TRID_LIB_PATH=$PWD/_ab/lib_prev.so python tools/eval_time.py rn50 128 512 --only-p16 2>&1 | grep "P16 eval" | sed 's/^/old /' | tee -a $O/eval.txt
done
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
  TRID_LIB_PATH=$PWD/_ab/lib_prev.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
done
