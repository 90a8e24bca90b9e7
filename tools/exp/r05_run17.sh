#!/bin/bash
# round 5, run 17: full GPU suite + the bench line with default flags
mkdir -p gpurun_out/r05_run17
O=gpurun_out/r05_run17
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['step_launch'][:40], d['config']['launch_probe']); print(d['retrieval']['value'], d['retrieval']['roofline'].get('rocprof_kernel')); print(d['gallery_encode']['by_batch'])"
