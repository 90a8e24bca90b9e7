#!/bin/bash
# round 5, run 14: the recorded step replayed as stream launches (csrc/step_replay.hip): capture tests, then the step with driver flags
mkdir -p gpurun_out/r05_run14
O=gpurun_out/r05_run14
python -m pytest tests/test_match_state_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "captur or graph or replay or determin" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-retrieval --no-configs3 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.2f  host %.1f' % (d['ms_per_step'], d['config']['host_enqueue_ms_per_step']), d['config']['launch_probe'])" | tee -a $O/ab.txt
grep "launch probe\|captured" $O/bench.err
done
