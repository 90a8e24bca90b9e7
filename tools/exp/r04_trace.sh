#!/bin/bash
# per-queue busy time / concurrency of two replayed steps of the current build -> gpurun_out/trace_<tag>.txt
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04}
OUT="$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > "$OUT/line.json" 2> "$OUT/kt.err" || echo "rocprof failed" >&2
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
cd "$GRAFT_REPO_ROOT"
python tools/trace_streams.py "$f" adam 9 11 > "$OUT/streams.txt" 2>&1
python tools/trace_busy.py "$f" 0.3 > "$OUT/busy.txt" 2>&1
python tools/trace_chain.py "$f" 9 10 > "$OUT/chain.txt" 2>&1
python tools/trace_shapes.py "$f" 9 11 > "$OUT/shapes.txt" 2>&1
rm -rf "$OUT/kt"
cat "$OUT/streams.txt" | head -80
