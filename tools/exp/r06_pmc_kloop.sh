#!/bin/bash
# GPU box: SQ counter breakdown of the 3x3 P16 GEMM main loops (variant 3 = barrier-per-tile loop, 12 = software pipeline) on
# the layer4 shape (M = 24576, N = 512, K = 4608), random operands -> gpurun_out/<tag>/pmc_kloop.txt
TAG=${1:-r06b}; shift
VARIANTS=${@:-3 12}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_kloop.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
: > $OUT
for v in $VARIANTS; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/tools/kloop_one.py 24 8 512 $v 0 4 > /dev/null 2>/tmp/pm.err
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "## variant $v counters: $set" >> $OUT
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $f "gemm_p16_kernel" >> $OUT; else tail -3 /tmp/pm.err >> $OUT; fi
done
done
cat $OUT
