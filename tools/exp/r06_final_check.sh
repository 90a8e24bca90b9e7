#!/bin/bash
# what the driver runs at round end, on HEAD: the GPU suite, smoke(), the bench line
O=gpurun_out/r06_final_check; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('rocprof_one_lane',{}).get('frac'), d.get('replay_equals_eager_b128'), d['parity_vs_oracle']['worst_rel_err'], d['configs3_1gpu']['bf16_speedup'])"
