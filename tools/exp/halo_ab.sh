#!/bin/bash
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do for L in ab/lib_nofd.so ab/lib_fd.so; do echo "== $L"; TRID_LIB_PATH=$L python tools/stem_bench.py 2>&1 | grep "ring-of-rows kernel, with"; done; done
