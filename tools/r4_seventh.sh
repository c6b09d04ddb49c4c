#!/bin/bash
# round 4, GPU call 7: whole GPU suite after fp16x3 / predict graph, then latency
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r4_tests5.log 2>&1; rc=$?
grep -E "fp16x3|passed|failed|FAILED|Error" gpurun_out/r4_tests5.log | tail -30
[ $rc -eq 0 ] || tail -80 gpurun_out/r4_tests5.log
python tools/latency_b1.py bf16x3,fp16x3,fp16 > gpurun_out/r4_latency3.log 2>&1; grep "^L=" gpurun_out/r4_latency3.log
