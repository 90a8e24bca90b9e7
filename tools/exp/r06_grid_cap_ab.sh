#!/bin/bash
# GPU box: cap on the grids of the elementwise (BatchNorm apply / backward apply / pool) passes - 4096 workgroups (every wave slot
# of the chip, two rounds) against fewer resident waves that leave room for a GEMM workgroup of another lane on the same CU -
# the passes alone (tools/bn_p16_bench.py) and in the step, same box -> gpurun_out/<tag>/grid_cap_ab.txt
TAG=${1:-r06ah}
O=gpurun_out/$TAG; mkdir -p $O; : > $O/grid_cap_ab.txt
for cap in 4096 2048 1024 512; do
  echo "== cap $cap, passes alone" >> $O/grid_cap_ab.txt
  TRID_GRID_CAP=$cap timeout 300 python tools/bn_p16_bench.py 2>/dev/null | grep -E "weighted total" >> $O/grid_cap_ab.txt
done
for rep in 1 2; do
for cap in 4096 2048 1024 512; do
  TRID_GRID_CAP=$cap TRID_BENCH_LAUNCH=streams timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/b.json 2> $O/b.err
  python -c "
import json; d=json.load(open('$O/b.json')); print('cap $cap step', d['ms_per_step'], d.get('replay_equals_eager_b128'))" >> $O/grid_cap_ab.txt
done
done
cat $O/grid_cap_ab.txt
