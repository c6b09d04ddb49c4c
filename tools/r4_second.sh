#!/bin/bash
# round 4, GPU call 2: whole GPU suite (fp16 operand format, new API pieces) + bench + single-frame latency
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r4_tests1.log 2>&1; rc=$?
grep -E "fp16 |^outliers|batch 12|bf16 step|adam traj|passed|failed|FAILED|Error" gpurun_out/r4_tests1.log | tail -40
[ $rc -eq 0 ] || { tail -60 gpurun_out/r4_tests1.log; exit 1; }
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench1.log 2>&1 && tail -1 gpurun_out/r4_bench1.log > gpurun_out/r4_bench1.json && python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench1.json'))
print('value', d['value'], d['dtype'], 'parity', d['parity'])
print('bf16_mode', d.get('bf16_mode'))
print('parity_mode', d.get('parity_mode'))
print('roofline', d['roofline']['achieved'], d['roofline']['avg_launch_ms'], 'kernels', d['kernel_ms_per_step'], 'one_stream', d.get('one_stream'))
PY
python tools/latency_b1.py > gpurun_out/r4_latency.log 2>&1 && cat gpurun_out/r4_latency.log
