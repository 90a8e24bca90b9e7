#!/bin/bash
# A/B of an environment switch on one box: tools/exp/r04_run10.sh VAR  (VAR=0 vs VAR=1, two rounds)
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
V=$1
for i in 1 2; do
  for v in 0 1; do
    env $V=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$v ms_per_step %.2f' % d['ms_per_step'])"
  done
done
