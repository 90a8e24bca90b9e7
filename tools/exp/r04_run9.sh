#!/bin/bash
# same-box A/B: four streams vs one stream (TRID_SERIAL=1), hipGraph replay both
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04k
run() {  # label, env...
  local L=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2> gpurun_out/r04k/$L.err > gpurun_out/r04k/$L.json || echo "$L failed"
  python - "$L" <<'PY'
import json,sys
L=sys.argv[1]
try:
    d=json.load(open("gpurun_out/r04k/%s.json"%L)); print(L, "ms_per_step %.2f"%d["ms_per_step"])
except Exception as e: print(L, "no line", e)
PY
}
run streams4_a TRID_X=0
run serial_a TRID_SERIAL=1
run streams4_b TRID_X=0
run serial_b TRID_SERIAL=1
run serial_wgrad_only TRID_SERIAL_WGRAD=1
run serial_k_only TRID_SERIAL_K=1
