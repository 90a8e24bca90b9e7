#!/bin/bash
# GPU box: the fused queue block with the compiler's own schedule (TRID_QNCE_SCHED=0) against the pinned one (1), same box,
# alternating -> gpurun_out/<tag>/qsim_sched_ab.txt
TAG=${1:-r06z}
OUT=gpurun_out/$TAG/qsim_sched_ab.txt
mkdir -p gpurun_out/$TAG; : > $OUT
for rep in 1 2; do
for s in 0 1; do
  for K in 8192 65536; do
    echo -n "sched $s " >> $OUT; TRID_QNCE_SCHED=$s timeout 300 python tools/qsim_one.py $K 200 2>/dev/null | tail -1 >> $OUT
  done
done
done
timeout 600 python -m pytest tests -m gpu -x -q -k "queue or infonce or nce" 2>&1 | tail -3 >> $OUT
cat $OUT
