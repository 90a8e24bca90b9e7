#!/bin/bash
# round 5, run 19: bn3 + downsample BatchNorm backward in one reduce / one apply pass: kernel test, model tests, step A/B
mkdir -p gpurun_out/r05_run19
O=gpurun_out/r05_run19
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "two_layers or bn_b" > $O/t1.txt 2>&1; tail -3 $O/t1.txt
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "full or config1 or block or step" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
for i in 1 2; do for v in 1 0; do
  TRID_BN_DUAL=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_BN_DUAL=$v ms_per_step %.2f (%s)' % (d['ms_per_step'], d['config']['launch_probe']['chosen']))" | tee -a $O/ab.txt
done; done
