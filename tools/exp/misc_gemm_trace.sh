#!/bin/bash
# which small GEMMs run between the last residual block and the losses (attention pool, head, losses), and for how long
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/misc_trace"; rm -rf "$OUT"; mkdir -p "$OUT"
TRID_CAPTURE=0 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-retrieval --no-configs3 > "$OUT/line.json" 2> "$OUT/err"
F=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
if [ -z "$F" ]; then echo "no trace" >&2; tail -5 "$OUT/err"; exit 1; fi
python3 - "$F" > "$OUT/misc.txt" <<'PY'
import csv, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","?"), r.get("Grid_Size_X", r.get("Grid_Size","?")), r.get("Workgroup_Size_X", "?")))
rows.sort()
ad=[i for i,r in enumerate(rows) if "adam_multi" in r[2]]
a,b=ad[-3],ad[-2]
sel=rows[a+1:b+1]
t0=sel[0][0]
print("step %.2f ms, %d kernels"%((sel[-1][1]-t0)/1e6, len(sel)))
big=("gemm_p16","bn_","conv3x3_halo","gru_","slab_reduce","p16_pack","stem_conv1","amax")
tot=collections.Counter(); cnt=collections.Counter()
for s,e,n,q,g,w in sel:
    short=n.split("(")[0].replace("void trid::","").replace("trid::","")[:70]
    if not any(x in n for x in big):
        print("%8.3f ms  +%7.1f us  q%s grid %s  %s"%((s-t0)/1e6,(e-s)/1e3,q,g,short))
    tot[short]+=e-s; cnt[short]+=1
print("---- totals")
for k,v in tot.most_common(60): print("%8.1f us  x%3d  %s"%(v/1e3,cnt[k],k))
PY
rm -rf "$OUT/kt"
cat "$OUT/misc.txt" | head -150
