#!/bin/bash
# r06r: kernel breakdown of ONE retrieval shard (125 000 gallery rows, 1e4 queries): where do the fixed costs go?  + the DP tests at HEAD
O=$GRAFT_REPO_ROOT/gpurun_out/r06r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TRID_RETR_ONLY_P16=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rt -- python3 $GRAFT_REPO_ROOT/tools/retrieval_time.py 125000 > $O/retrieval_shard_time.txt 2> $O/rt.err
cd $GRAFT_REPO_ROOT
f=$(find $O/rt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/retrieval_shard_kernel_stats.csv
f=$(find $O/rt -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 - "$f" > $O/retrieval_shard_trace_tail.txt <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last call of similarity_topk: kernels after the last big gap
last=rows[-40:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("%9.1f us  +%7.1f us  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:90]))
P
rm -rf $O/rt
cat $O/retrieval_shard_time.txt | tail -2; head -14 $O/retrieval_shard_kernel_stats.csv | cut -c1-160
timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q -m gpu -k "not eight_ranks and not four_ranks" > $O/dp_tests.txt 2>&1; tail -2 $O/dp_tests.txt
