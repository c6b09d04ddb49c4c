#!/bin/bash
set -e
python -m pytest tests/test_fp16_gpu.py -q -x -m gpu 2>&1 | tail -2
for i in 1 2; do
python bench.py --config parity --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('parity fp16x3', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogp'], d['parity']['argmax_flips'])"
DINOSEG_LIB=$GRAFT_REPO_ROOT/build/variants/lib_prev.so python bench.py --config parity --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('parity fp16x3 (previous build)', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogp'])"
done
