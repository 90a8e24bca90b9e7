#!/bin/bash
# same-box before / after of round 4: the round-3 tree (built beside the current one:
#   mkdir old_tree && git archive 954b2ac | tar -x -C old_tree && make -C old_tree/textreid_amd/csrc
# - not committed) against the current tree, three alternating default-path bench runs each
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  (cd old_tree && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round-3 build: %.2f ms/step  %.0f pairs/s' % (d['ms_per_step'], d['value']))")
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round-4 build: %.2f ms/step  %.0f pairs/s  (%s; probe %s)' % (d['ms_per_step'], d['value'], d['config']['step_launch'][:14], {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['config']['launch_probe'].items()}))"
done
