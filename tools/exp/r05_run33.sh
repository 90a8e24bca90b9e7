#!/bin/bash
# round 5, run 33: grid shapes of the step-boundary multi-tensor kernels (Adam / EMA chunk, amax slices, pack workgroups per tensor)
mkdir -p gpurun_out/r05_run33
O=gpurun_out/r05_run33/ab.txt; : > $O
run() { TRID_BENCH_LAUNCH=streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 ms_per_step %.2f' % d['ms_per_step'])" | tee -a $O; }
for i in 1 2; do
  run base
  TRID_ADAM_CHUNK=16384 TRID_EMA_CHUNK=16384 run chunk16k
  TRID_AMAX_GY=64 TRID_PACK_GX=128 run amax64_pack128
  TRID_ADAM_CHUNK=16384 TRID_EMA_CHUNK=16384 TRID_AMAX_GY=64 TRID_PACK_GX=128 run all
done
