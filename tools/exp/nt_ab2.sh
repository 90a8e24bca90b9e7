#!/bin/bash
# non-temporal accesses in the streaming GEMM / ring-of-rows kernels and the BatchNorm passes (runtime switches), same box
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/nt_ab2; mkdir -p $OUT
for v in 0 1 2 3; do echo "== TRID_STREAM_NT=$v (bit 0: activation loads, bit 1: output stores)"; TRID_STREAM_NT=$v timeout 300 python tools/stream_bench.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/stream.txt
for v in 0 1; do echo "== TRID_STREAM_NT=$v"; TRID_STREAM_NT=$v timeout 300 python tools/stem_bench.py 2>&1 | grep -E "ring-of-rows"; done | tee $OUT/stem.txt
for i in 1 2; do for cfg in "TRID_BN_NT=0 TRID_STREAM_NT=0" "TRID_BN_NT=1 TRID_STREAM_NT=0" "TRID_X=0" ; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$cfg]', round(d['ms_per_step'],2))"
done; done | tee $OUT/step.txt
