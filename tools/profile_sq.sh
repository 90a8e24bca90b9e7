#!/bin/bash
# GPU box: SQ counter breakdown (issue / LDS / wait / MFMA) of the split GEMM on the big 3x3 shapes and of the queue kernel
# -> gpurun_out/prof_sq_<tag>.txt   usage: tools/profile_sq.sh <tag>
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_sq_$TAG.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
export TRID_LB_PREC=16 TRID_LB_FILTER="l3.0.conv2,l4.x.conv2,l4.x.conv3"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/tools/layer_bench.py > /dev/null 2>/tmp/pm.err
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "## counters: $set" >> $OUT
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $f "gemm_bf16s_kernel<2, 0, 16, 128>" >> $OUT; python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $f "gemm_bf16s_kernel<0, 0, 16, 128>" >> $OUT; else tail -3 /tmp/pm.err >> $OUT; fi
done
cat $OUT
