#!/bin/bash
# SQ counter breakdown of the P16 GEMM variants -> gpurun_out/prof_p16_<tag>.txt   usage: tools/p16_prof.sh <tag> <variants...>
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
TAG=${1:?usage: tools/p16_prof.sh <tag> <variants...>}; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_p16_$TAG.txt"
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out"
PM=$(mktemp -d /tmp/pm.XXXXXX)
cd /tmp && export TMPDIR=/tmp
: > "$OUT"
for v in "$@"; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT"; do
  rm -rf "$PM"; rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$PM" -- python3 "$GRAFT_REPO_ROOT/tools/p16_prof.py" $v > /dev/null 2>"$PM.err" || true
  f=$(find "$PM" -name "*counter_collection.csv" 2>/dev/null | head -1)
  echo "## variant $v counters: $set" >> "$OUT"
  if [ -n "$f" ]; then python3 $GRAFT_REPO_ROOT/tools/pmc_sq.py $f "gemm_p16_kernel" >> "$OUT"; else tail -3 "$PM.err" >> "$OUT"; fi
done
done
