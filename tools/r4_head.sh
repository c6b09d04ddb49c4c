#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -k "head_final" -m gpu 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_fp16_gpu.py -q -x -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-parity-mode --steps 30 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['parity'])" | tee gpurun_out/r4_head.log
