#!/bin/bash
# the last 2.5 ms before the optimizer launch of an EAGER step under rocprofv3 (tools/trace_tail.py): what the end of backward waits for
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tail; mkdir -p $OUT
TRID_BENCH_LAUNCH=eager rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/line.json 2> $OUT/err.txt
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT
python tools/trace_tail.py $f 22 2.5 > $OUT/tail.txt 2>&1
python tools/trace_streams.py $f adam 21 23 > $OUT/streams.txt 2>&1
rm -rf $OUT/kt
tail -45 $OUT/tail.txt; head -8 $OUT/streams.txt
