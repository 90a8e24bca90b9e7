#!/bin/bash
# round 5, run 21: P16 stores of the elementwise passes as whole lines (lane pairs swap plane halves): tests, bandwidth probe, step A/B
mkdir -p gpurun_out/r05_run21
O=gpurun_out/r05_run21
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bn or p16 or pool or block" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
python tools/exp/bn_bw_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/bn_bw.txt
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
  TRID_LIB_PATH=$PWD/_ab/lib_prev.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
done
