#!/bin/bash
# round 6 evidence set of one build: full GPU suite, rocprofv3 (four lanes, one lane, PMC passes; inference half), the bench line
# of the driver's command, the data-parallel host path (one-rank nccl group, TRID_DP_FORCE=1) segmented vs eager.  usage: r06_final.sh <tag>
TAG=${1:-r06j}
O=gpurun_out/${TAG}_final; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 900 bash tools/profile_round.sh $TAG > /dev/null 2>&1
timeout 400 bash tools/exp/r05_prof_one_lane.sh $TAG > /dev/null 2>&1
timeout 500 bash tools/exp/r05_prof_infer.sh $TAG > /dev/null 2>&1
timeout 700 bash tools/exp/r05_pmc_infer.sh $TAG > /dev/null 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['launch_probe']['chosen'], d['config'].get('host_work_ms_per_step_no_backpressure')); print(d['roofline']['frac'], d['roofline'].get('frac_isolated'), d['roofline'].get('frac_isolated_zero_operands'), d['roofline_second']['frac']); print(d['retrieval']['value'], d['retrieval'].get('one_of_8_shards')); print(d['gallery_encode']['by_batch']); print(d['parity_vs_oracle']['worst_rel_err'], d.get('replay_equals_eager_b128'))"
TRID_DP_FORCE=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_segmented.json 2> $O/bench_dp1_segmented.err
TRID_DP_FORCE=1 TRID_DP_CAPTURE=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_dp1_eager.json 2> $O/bench_dp1_eager.err
for f in dp1_segmented dp1_eager; do python -c "
import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config'].get('host_work_ms_per_step_no_backpressure'), d['config']['step_launch'][:60])"; done
ls gpurun_out/prof_$TAG gpurun_out/prof_infer_$TAG gpurun_out/pmc_infer_$TAG
