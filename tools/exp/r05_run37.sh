#!/bin/bash
# round 5, run 37: Adam launched per residual layer behind the layer's gradient events (TRID_ADAM_STAGED): same parameters after 5 steps?  step A/B
mkdir -p gpurun_out/r05_run37
O=gpurun_out/r05_run37
for v in 0 1; do echo "TRID_ADAM_STAGED=$v" >> $O/digest.txt; TRID_ADAM_STAGED=$v python tools/exp/replay_probe.py 128 2>&1 | grep "digest\|ok\|FAILED" | tail -3 >> $O/digest.txt; done
cat $O/digest.txt
TRID_ADAM_STAGED=1 python -m pytest tests/test_match_state_gpu.py -x -q -m gpu -k "captur or determin or do_train or replay" 2>&1 | tail -2
for i in 1 2 3; do for v in 1 0; do
  TRID_BENCH_LAUNCH=streams TRID_ADAM_STAGED=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRID_ADAM_STAGED=$v ms_per_step %.2f lanes %s events %s' % (d['ms_per_step'], d['config']['launch_probe']['stream_replay_plan']['lanes'], d['config']['launch_probe']['stream_replay_plan']['events']))" | tee -a $O/ab.txt
done; done
