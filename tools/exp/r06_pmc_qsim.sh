#!/bin/bash
# GPU box: SQ counter breakdown of the fused queue block's main kernel (queue_nce.hip) at K = 8192 and K = 65536
# -> gpurun_out/<tag>/pmc_qsim.txt
TAG=${1:-r06y}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_qsim.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
: > $OUT
for K in 8192 65536; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pm; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/tools/qsim_one.py $K 8 > /dev/null 2>/tmp/pm.err
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "## K $K counters: $set" >> $OUT
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $f "queue_nce_f16_kernel" >> $OUT; else tail -3 /tmp/pm.err >> $OUT; fi
done
done
cat $OUT
