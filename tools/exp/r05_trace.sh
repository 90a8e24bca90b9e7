#!/bin/bash
# GPU box: kernel timeline of the stream-replayed step (rocprofv3 --kernel-trace, no stats): idle gaps of the chip and of the main queue
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/r05_trace"; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TRID_BENCH_LAUNCH=streams rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/line.json 2> $OUT/kt.err
cd $GRAFT_REPO_ROOT
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 27 31 > $OUT/gaps.txt 2>&1
python tools/trace_streams.py $f adam 28 30 > $OUT/streams.txt 2>&1 || true
python tools/trace_busy.py $f 0.3 > $OUT/busy.txt 2>&1 || true
python tools/trace_lane_gaps.py $f 27 30 > $OUT/lane_gaps.txt 2>&1 || true
python tools/trace_queue.py $f 28 4 24.0 30.0 > $OUT/queue4.txt 2>&1 || true
rm -rf $OUT/kt
cat $OUT/queue4.txt | cut -c1-120
