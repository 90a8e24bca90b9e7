cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c3prof; rm -rf $OUT; mkdir -p $OUT
for g in 0 1; do
export TRID_BF16_GRADS=$g
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks$g -- python3 $GRAFT_REPO_ROOT/tools/exp/c3_prof.py 1 > $OUT/log$g 2>&1
cp $(find $OUT/ks$g -name "*kernel_stats.csv" | head -1) $OUT/stats_bf16_g$g.csv; rm -rf $OUT/ks$g
done
