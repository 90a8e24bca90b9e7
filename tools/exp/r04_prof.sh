#!/bin/bash
# kernel-trace stats of the current build -> gpurun_out/prof_<tag>/
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04}
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > "$OUT/bench_line_profiled.json" 2> "$OUT/ks.err" || echo "rocprof failed" >&2
f=$(find "$OUT/ks" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/bench_kernel_stats.csv"; fi
rm -rf "$OUT/ks"
cd "$GRAFT_REPO_ROOT" && python tools/kernel_stats_per_step.py "$OUT/bench_kernel_stats.csv" 19 45
