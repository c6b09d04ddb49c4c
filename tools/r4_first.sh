#!/bin/bash
# round 4, first GPU call: baseline tests + attention row-sum variants (AZ_ROWSUM) interleaved + a short bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4_tests0.log 2>&1 || { tail -20 gpurun_out/r4_tests0.log; exit 1; }
tail -3 gpurun_out/r4_tests0.log
for v in rs1 rs2; do
  DINOSEG_LIB=build/variants/lib_$v.so python -m pytest tests/test_ops_gpu.py -q -k attention > gpurun_out/r4_attn_$v.log 2>&1 || { tail -20 gpurun_out/r4_attn_$v.log; exit 1; }
  tail -1 gpurun_out/r4_attn_$v.log
done
bash tools/ab_ops.sh attn 3 cur build/variants/lib_rs1.so build/variants/lib_rs2.so > gpurun_out/r4_ab_attn.log 2>&1 && cat gpurun_out/r4_ab_attn.log | grep "planes=1" &&
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench0.log 2>&1 && tail -1 gpurun_out/r4_bench0.log | cut -c1-600
