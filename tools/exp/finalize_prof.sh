#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/finprof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 "$GRAFT_REPO_ROOT/tools/exp/finalize_bench.py" > "$OUT/log.txt" 2>&1
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
c=collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if "finalize" not in n: continue
    k=(n[:60], r["Grid_Size_X"])
    v=c.setdefault(k,[0,0.0]); v[0]+=1; v[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for (n,g),(cnt,us) in c.items(): print("%-62s grid %7s x%-3d avg %6.1f us"%(n,g,cnt,us/cnt))
PY
rm -rf "$OUT/kt"
