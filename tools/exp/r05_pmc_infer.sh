#!/bin/bash
# GPU box: PMC passes of the inference half (separate --pmc runs, --kernel-trace only): HBM-side traffic and MFMA-busy of the
# retrieval filter kernel and of the eval-mode encoder's kernels -> gpurun_out/pmc_infer_<tag>/
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r05}
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_infer_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TRID_RETR_ONLY_P16=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/rt_$c -- python3 $GRAFT_REPO_ROOT/tools/retrieval_time.py > /dev/null 2> $OUT/rt_$c.err
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/ev_$c -- python3 $GRAFT_REPO_ROOT/tools/eval_time.py rn50 128 --only-p16 > /dev/null 2> $OUT/ev_$c.err
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/rt_mfma -- python3 $GRAFT_REPO_ROOT/tools/retrieval_time.py > /dev/null 2> $OUT/rt_mfma.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/ev_mfma -- python3 $GRAFT_REPO_ROOT/tools/eval_time.py rn50 128 --only-p16 > /dev/null 2> $OUT/ev_mfma.err
cd $GRAFT_REPO_ROOT
for k in rt ev; do
  python tools/pmc_summary.py $(find $OUT/${k}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/${k}_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $OUT/${k}_pmc_hbm_traffic.txt 2>&1
  python tools/pmc_mfma_util.py $(find $OUT/${k}_mfma -name "*counter_collection.csv" | head -1) > $OUT/${k}_pmc_mfma_util.txt 2>&1
done
rm -rf $OUT/rt_FETCH_SIZE $OUT/rt_WRITE_SIZE $OUT/ev_FETCH_SIZE $OUT/ev_WRITE_SIZE $OUT/rt_mfma $OUT/ev_mfma
head -6 $OUT/rt_pmc_hbm_traffic.txt; head -5 $OUT/rt_pmc_mfma_util.txt; head -12 $OUT/ev_pmc_hbm_traffic.txt; head -10 $OUT/ev_pmc_mfma_util.txt
