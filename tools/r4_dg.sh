#!/bin/bash
set -e
python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -q -x -m gpu 2>&1 | tail -2
for i in 1 2; do python bench.py --config finetune --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('finetune', d['value'], d['ms_per_step'], d['gradient_parity']['worst_sampled_relative_error'])"; done
export DINOSEG_LIB=$GRAFT_REPO_ROOT/build/variants/lib_prev.so
for i in 1 2; do python bench.py --config finetune --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('finetune (previous build)', d['value'], d['ms_per_step'])"; done
