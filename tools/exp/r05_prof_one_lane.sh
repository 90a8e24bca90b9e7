#!/bin/bash
# GPU box: rocprofv3 kernel stats of the replayed step laid out on ONE stream (TRID_STEP_LANES=1): every kernel has the chip to
# itself, its begin-to-end duration is its own -> gpurun_out/prof_<tag>/bench_kernel_stats_one_lane.csv
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r05g}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TRID_STEP_LANES=1 TRID_BENCH_LAUNCH=streams
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench_line_one_lane.json 2> $OUT/k1.err
cd $GRAFT_REPO_ROOT
f=$(find "$OUT/k1" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_kernel_stats_one_lane.csv"
rm -rf "$OUT/k1"
head -6 $OUT/bench_kernel_stats_one_lane.csv | cut -c1-160
python3 -c "import json; d=json.load(open('$OUT/bench_line_one_lane.json')); print(d['ms_per_step'], d['config']['launch_probe'])"
