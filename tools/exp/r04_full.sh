#!/bin/bash
# the driver's GPU tier: the whole -m gpu suite + smoke, then an A/B of this round's switches on the same box
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_full; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench${rep}_new.json 2> $OUT/bench${rep}_new.err
TRID_P16_STEM=0 TRID_STREAM_1X1=0 TRID_SKINNY_GEMM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 > $OUT/bench${rep}_old.json 2> $OUT/bench${rep}_old.err
done
python - <<'PY'
import json
for f in ("bench1_new","bench1_old","bench2_new","bench2_old"):
    try:
        d=json.load(open("gpurun_out/r04_full/%s.json"%f)); print(f, "ms_per_step %.2f"%d["ms_per_step"])
    except Exception as e: print(f, "failed", e)
PY
