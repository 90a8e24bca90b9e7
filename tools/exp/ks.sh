cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_tmp; rm -rf $OUT; mkdir -p $OUT
TRID_BN_FINALIZE_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/line.json 2> $OUT/err
cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/stats.csv; rm -rf $OUT/ks
grep -E "bn_finalize|bn_bwd_reduce_final" $OUT/stats.csv | cut -c1-60,180-
