#!/bin/bash
# round 4, GPU call 6: fp16 hi+lo mode (fp16x3): tests + bench records
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_fp16_gpu.py tests/test_model_gpu.py -m gpu -q -s -k "fp16 or two_stream_split_equals_one_stream_at" > gpurun_out/r4_tests4.log 2>&1; rc=$?
grep -E "fp16x3|passed|failed|FAILED|Error" gpurun_out/r4_tests4.log | tail -30
[ $rc -eq 0 ] || tail -60 gpurun_out/r4_tests4.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench2.log 2>&1 && tail -1 gpurun_out/r4_bench2.log > gpurun_out/r4_bench2.json && python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench2.json'))
print('value', d['value'], d['dtype'], 'parity', d['parity'])
for k in ('bf16_mode', 'parity_mode', 'parity_mode_bf16x3'):
    print(k, d.get(k))
PY
