#!/bin/bash
# GPU box: SQ counter breakdown (issue / LDS / wait / MFMA) of the retrieval filter kernel -> gpurun_out/sq_retrieval.txt
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_retrieval.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT
export TRID_RETR_ONLY_P16=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/tools/retrieval_time.py 200000 > /dev/null 2>/tmp/pm.err
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  echo "## counters: $set" >> $OUT
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $f "gemm_p16_stream_kernel<256, 8, 2, false, 4>" >> $OUT; else tail -3 /tmp/pm.err >> $OUT; fi
done
cat $OUT
