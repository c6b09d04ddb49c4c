#!/bin/bash
# GPU box: the hi + lo fused launch (mlp_fused3.hip): op tests, the launch alone, and the parity-mode line with / without the qkv tail
OUT=gpurun_out/${1:-m3d}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "hi_lo_planes" -s > $OUT/test.log 2>&1
grep -E "tail3|passed|failed|Error|assert" $OUT/test.log | tail -30
timeout -k 10 200 python tools/bench_mlp3.py 115232 20 1 1 2>&1 | tail -1
for opt in "" "--option qkv_fused3=0"; do
  timeout -k 10 300 python bench.py --config parity --no-cpu-baseline $opt > $OUT/parity.log 2>&1
  python -c "
import json; d=json.loads(open('$OUT/parity.log').read().strip().splitlines()[-1]); print('$opt', d['value'], d['one_stream']['value'], d['parity'], d['kernel_ms_per_step'])"
done
