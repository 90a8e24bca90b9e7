#!/bin/bash
# r06d: the GPU test suite on the software-pipelined default; bench.py A/B on one box: default (variant 12), weight gradients with
# the SP loop too, and the round-5 loop (variant 3)
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval > $O/bench_v12.json 2> $O/bench_v12.log
TRID_WGRAD_SP=1 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_v12_wsp.json 2> $O/bench_v12_wsp.log
TRID_P16_DEFAULT_VARIANT=3 python bench.py --steps 20 --warmup 5 --no-configs3 --no-retrieval --no-cpu-baseline > $O/bench_v3.json 2> $O/bench_v3.log
for f in v12 v12_wsp v3; do python - $O/bench_$f.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], "seam", d.get("replay_equals_eager_b128"), "roof", d["roofline"]["frac"], d["roofline"].get("frac_isolated"), d["roofline"].get("frac_isolated_zero_operands"))
P
done
