#!/bin/bash
# non-temporal loads / stores in the BatchNorm passes (build with -DTRID_STREAM_NT): per-pass bandwidth and the whole step, A/B on one box
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/nt_ab; mkdir -p $OUT
for L in textreid_amd/libtextreid_hip.so tools/exp/libtextreid_nt.so; do
  echo "== $L"; TRID_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout 300 python tools/bn_bench.py 2>&1 | grep -v amdgpu.ids
done | tee $OUT/bn_bench.txt
for i in 1 2; do for L in textreid_amd/libtextreid_hip.so tools/exp/libtextreid_nt.so; do
  TRID_LIB_PATH=$GRAFT_REPO_ROOT/$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$L', round(d['ms_per_step'],2))"
done; done | tee $OUT/step.txt
