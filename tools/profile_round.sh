#!/bin/bash
# GPU box: rocprofv3 evidence of the current build -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/)
# usage: tools/profile_round.sh <tag>
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- $BENCH > $OUT/bench_line_profiled.json 2> $OUT/ks.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/qs -- python3 $GRAFT_REPO_ROOT/tools/qsim_prof.py > $OUT/qsim.log 2>&1
BENCH2="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-retrieval --no-configs3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH2 > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH2 > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $BENCH2 > /dev/null 2> $OUT/mfma.err
cd $GRAFT_REPO_ROOT
f=$(find "$OUT/ks" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_kernel_stats.csv"
f=$(find "$OUT/qs" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/qsim_kernel_stats.csv"
python tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) > $OUT/pmc_hbm_traffic.txt 2>&1
python tools/pmc_mfma_util.py $(find $OUT/mfma -name "*counter_collection.csv" | head -1) > $OUT/pmc_mfma_util.txt 2>&1
rm -rf "$OUT/ks" "$OUT/qs" "$OUT/fetch" "$OUT/write" "$OUT/mfma"
ls -la "$OUT"
