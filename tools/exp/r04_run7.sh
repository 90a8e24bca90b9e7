#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "skinny or gemm_nt or gemm_nn or arithmetic_modes or small_ops" > $OUT/t_k.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/t_k.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "visual_encoder or losses or moco_head or full_size_step_vs_oracle or eval_bn or odd_batch" > $OUT/t_m.log 2>&1; echo "model rc=$?"; tail -5 $OUT/t_m.log | cut -c1-300
for rep in 1 2; do for v in 1 0; do
TRID_SKINNY_GEMM=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench${rep}_sk$v.json 2> $OUT/bench${rep}_sk$v.err
done; done
python - <<'PY'
import json
for f in ("bench1_sk1","bench1_sk0","bench2_sk1","bench2_sk0"):
    try:
        d=json.load(open("gpurun_out/r04g/%s.json"%f)); print(f, "ms_per_step %.2f"%d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
