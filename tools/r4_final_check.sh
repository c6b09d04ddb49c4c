#!/bin/bash
# the driver's round-end sequence on one box: GPU tests, smoke(), the default bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_final_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4_final_tests.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r4_final_tests.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err; tail -1 gpurun_out/r4_final_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], d['dtype'], 'roofline', d['roofline']['frac'], d['roofline']['traffic'], 'parity', d['parity']['argmax_flips'], d['parity']['max_abs_dlogp'], 'cpu', d['cpu_baseline']['value'], 'bf16', d['bf16_mode']['value'], 'fp16x3', d['parity_mode']['value'], d['parity_mode']['parity']['max_abs_dlogp'])"
