#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_blocks_gpu.py -q -x -s -k "stem" > $OUT/t_stem.log 2>&1; echo "stem rc=$?"
tail -8 $OUT/t_stem.log | cut -c1-600
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -s -k "visual_encoder or full_size_step or config1 or full_batch" > $OUT/t_model.log 2>&1; echo "model rc=$?"
tail -8 $OUT/t_model.log | cut -c1-600
for v in 1 0; do
TRID_P16_STEM=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench_stem$v.json 2> $OUT/bench_stem$v.err; echo "bench stem=$v rc=$?"
done
for v in 1 0; do
TRID_P16_STEM=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench2_stem$v.json 2> $OUT/bench2_stem$v.err
done
python - <<'PY'
import json
for f in ("bench_stem1","bench_stem0","bench2_stem1","bench2_stem0"):
    try:
        d=json.load(open("gpurun_out/r04c/%s.json"%f)); print(f, "ms_per_step %.2f"%d["ms_per_step"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "failed", e)
PY
