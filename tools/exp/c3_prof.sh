cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c3prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- python3 $GRAFT_REPO_ROOT/tools/exp/c3_prof.py 1 > $OUT/log 2>&1
cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/stats_bf16.csv; rm -rf $OUT/ks
