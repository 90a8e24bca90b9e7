#!/bin/bash
# round 5, last build: full GPU suite + the bench line (the kernel-level rocprof files of r05i still describe this build's kernels)
mkdir -p gpurun_out/r05_final2
O=gpurun_out/r05_final2
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['launch_probe']); print(d['retrieval']['value']); print(d['gallery_encode']['by_batch']); print(d['parity_vs_oracle']['worst_rel_err'])"
