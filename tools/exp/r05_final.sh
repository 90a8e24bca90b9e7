#!/bin/bash
# round 5, final build: full GPU suite, rocprofv3 evidence (four lanes, one lane, PMC passes), the bench line
mkdir -p gpurun_out/r05_final
O=gpurun_out/r05_final
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
bash tools/profile_round.sh r05n > /dev/null 2>&1
bash tools/exp/r05_prof_one_lane.sh r05n > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['launch_probe']['chosen']); print(d['retrieval']['value']); print(d['gallery_encode']['by_batch']); print(d['parity_vs_oracle']['worst_rel_err'])"
ls gpurun_out/prof_r05n
