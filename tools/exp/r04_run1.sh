#!/bin/bash
# round 4, first GPU call: the new parity tests + a baseline bench line of the unchanged kernels
set -uo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04a; mkdir -p $OUT
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; rocm-smi --showmeminfo vram >> $OUT/host.txt 2>&1
python -m pytest tests/test_kernels_gpu.py -q -s -k "fused_queue_infonce or dynamic_range or tiny_variance" > $OUT/t_kernels.log 2>&1; echo "kernels rc=$?"
python -m pytest tests/test_match_state_gpu.py -q -s -k "captured or failed_capture or do_train" > $OUT/t_state.log 2>&1; echo "state rc=$?"
python -m pytest tests/test_model_gpu.py -q -s -k "config1_b128" > $OUT/t_b128.log 2>&1; echo "b128 rc=$?"
python -m pytest tests/test_dp_gpu.py -q -s -k "config2 or rccl" > $OUT/t_dp2.log 2>&1; echo "dp config2 rc=$?"
python -m pytest tests/test_dp_gpu.py -q -s -k "config3" > $OUT/t_dp3.log 2>&1; echo "dp config3 rc=$?"
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -3 $OUT/t_*.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04a/bench_line.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("parity", d.get("parity_vs_oracle"))
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k!="sample"})
PY
