#!/bin/bash
# usage: tools/ab_bench.sh libA.so libB.so [rounds]   -- alternating bench runs of two library builds on ONE box
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    TRID_LIB_PATH=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$L', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), round(d['roofline_second']['achieved'],1))"
  done
done
