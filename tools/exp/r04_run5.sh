#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_p16_stream" > $OUT/t_stream.log 2>&1; echo "stream rc=$?"
timeout 1500 python -m pytest tests/test_blocks_gpu.py -q -x -s -k "p16" > $OUT/t_blocks.log 2>&1; echo "blocks rc=$?"
grep -E "passed|failed|B=128" $OUT/t_blocks.log | cut -c1-260 | tail -12
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -s -k "full_size or config1 or config3_rn101_k65536_fp32" > $OUT/t_model.log 2>&1; echo "model rc=$?"
tail -4 $OUT/t_model.log | cut -c1-400
for rep in 1 2; do for v in 1 0; do
TRID_STREAM_1X1=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench${rep}_stream$v.json 2> $OUT/bench${rep}_stream$v.err
done; done
python - <<'PY'
import json
for f in ("bench1_stream1","bench1_stream0","bench2_stream1","bench2_stream0"):
    try:
        d=json.load(open("gpurun_out/r04e/%s.json"%f)); print(f, "ms_per_step %.2f"%d["ms_per_step"], "1x1 frac %.3f"%d["roofline_second"]["frac"], "avg1x1 ms %.4f"%d["roofline_second"]["avg_launch_ms"])
    except Exception as e: print(f, "failed", e)
PY
