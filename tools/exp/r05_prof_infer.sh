#!/bin/bash
# GPU box: rocprofv3 kernel stats of the inference half - the configs[4] match and the eval-mode gallery encode -> gpurun_out/prof_infer_<tag>/
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r05}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_infer_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TRID_RETR_ONLY_P16=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rt -- python3 $GRAFT_REPO_ROOT/tools/retrieval_time.py > $OUT/retrieval_time.txt 2> $OUT/rt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ev -- python3 $GRAFT_REPO_ROOT/tools/eval_time.py rn50 128 --only-p16 > $OUT/eval_time.txt 2> $OUT/ev.err
cd $GRAFT_REPO_ROOT
f=$(find "$OUT/rt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/retrieval_kernel_stats.csv"
f=$(find "$OUT/ev" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/eval_encode_kernel_stats.csv"
rm -rf "$OUT/rt" "$OUT/ev"
head -12 $OUT/retrieval_kernel_stats.csv | cut -c1-200
