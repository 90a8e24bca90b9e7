# rocprofv3 kernel stats of ten configs[3]-shape steps (RN101, K=65536, B=128) in the bf16 mode and the fp32-class default
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repo copy there)}"
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/c3prof"; rm -rf "$OUT"; mkdir -p "$OUT"
for prec in 1 16; do
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ks$prec" -- python3 "$GRAFT_REPO_ROOT/tools/exp/c3_prof.py" $prec > "$OUT/log$prec" 2>&1 || echo "rocprofv3 pass (precision $prec) failed: see $OUT/log$prec" >&2
f=$(find "$OUT/ks$prec" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/stats_prec$prec.csv"; else echo "no kernel_stats.csv for precision $prec" >&2; fi
rm -rf "$OUT/ks$prec"
done
