#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "skinny or gemm_batched or gemm_nt or gemm_nn" > $OUT/t_k.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/t_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "visual_encoder or moco_head or full_size_step_vs_oracle or odd_batch" > $OUT/t_m.log 2>&1; echo "model rc=$?"; tail -5 $OUT/t_m.log | cut -c1-300
for rep in 1 2; do for v in 192 100000; do
python - $v > $OUT/bench${rep}_k$v.json 2> $OUT/bench${rep}_k$v.err <<'PY'
import sys, runpy
from textreid_amd import ops
ops.SKINNY_MIN_K_BATCHED = int(sys.argv[1])
_skinny_ok0 = ops._skinny_ok
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-retrieval", "--no-configs3"]
runpy.run_path("bench.py", run_name="__main__")
PY
done; done
python - <<'PY'
import json
for f in ("bench1_k192","bench1_k100000","bench2_k192","bench2_k100000"):
    try:
        d=json.load(open("gpurun_out/r04h/%s.json"%f)); print(f, "ms_per_step %.2f"%d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
