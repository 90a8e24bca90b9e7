set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repo copy there)}"
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/trace_tmp"; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/line.json 2> $OUT/err
F=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
if [ -z "$F" ]; then echo "no kernel trace was written: see $OUT/err" >&2; exit 1; fi
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $F 6 9 > $OUT/gaps.txt 2>&1
python tools/trace_streams.py $F adam 6 9 > $OUT/streams.txt 2>&1
python - "$F" > $OUT/order.txt <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id","?")))
rows.sort()
ad=[r for r in rows if "adam_multi" in r[2]]
a,b=ad[7][1],ad[8][1]
sel=[r for r in rows if r[0]>=a and r[1]<=b]
# coarse timeline: per 1 ms bucket, busy time per queue
import collections
qs=sorted({r[3] for r in sel})
print("queues",qs, "step ms",(b-a)/1e6)
nb=int((b-a)/1e6)+1
for k in range(nb):
    lo=a+k*1e6; hi=lo+1e6
    busy=collections.Counter()
    names=collections.Counter()
    for s,e,n,q in sel:
        ov=min(e,hi)-max(s,lo)
        if ov>0: busy[q]+=ov; names[(q,n.split('(')[0][-38:])]+=ov
    top=[ "%s:%s"%(q,n) for (q,n),v in names.most_common(3)]
    print("%3d ms "%k+" ".join("%s=%3d%%"%(q,busy[q]/1e4) for q in qs)+"  "+" | ".join(top))
PY
rm -rf $OUT/kt
