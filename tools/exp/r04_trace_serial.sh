#!/bin/bash
# one-stream step (TRID_SERIAL=1): un-overlapped kernel durations per (kernel, grid) -> gpurun_out/trace_<tag>_serial/shapes.txt
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04}
OUT="$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}_serial"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TRID_SERIAL=1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > "$OUT/line.json" 2> "$OUT/kt.err" || echo "rocprof failed" >&2
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
cd "$GRAFT_REPO_ROOT"
python tools/trace_shapes.py "$f" 9 11 0.03 > "$OUT/shapes.txt" 2>&1
python tools/trace_streams.py "$f" adam 9 11 > "$OUT/streams.txt" 2>&1
rm -rf "$OUT/kt"
head -3 "$OUT/streams.txt"
