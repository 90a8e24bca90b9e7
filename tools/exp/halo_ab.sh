#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
A=${1:-ab/lib_fd.so}; B=${2:-ab/lib_sb.so}
for i in 1 2; do for L in $A $B; do echo "== $L"; TRID_LIB_PATH=$L python tools/stem_bench.py 2>&1 | grep "ring-of-rows kernel, with"; TRID_LIB_PATH=$L python tools/exp/halo64_bench.py 2>&1 | grep "^halo with" | head -1; done; done
