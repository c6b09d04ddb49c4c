#!/bin/bash
# round 4, GPU call 5: fp16 model tests after the dispatch change, latency breakdown, 2-coefficient GELU A/B
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -m gpu -q -k "outlier or dispatch or two_stream" > gpurun_out/r4_tests3.log 2>&1; rc=$?
tail -3 gpurun_out/r4_tests3.log
[ $rc -eq 0 ] || { tail -50 gpurun_out/r4_tests3.log; exit 1; }
python tools/latency_b1.py bf16x3,fp16,bf16 > gpurun_out/r4_latency2.log 2>&1; grep "^L=" gpurun_out/r4_latency2.log
DINOSEG_LIB=build/variants/lib_gelu2.so python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -m gpu -q -k "mlp_fused or proj_mlp or block_tail" 2>&1 | tail -2
for i in 1 2 3; do
  for lib in cur gelu2; do
    if [ $lib = cur ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=build/variants/lib_$lib.so; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', d['value'], 'one_stream', d['one_stream']['value'], 'qkv', k['qkv_gemm'], 'attn', k['attention'], 'mlp', k['fc1_gemm'], 'parity', d['parity']['argmax_flips'], d['parity']['max_abs_dlogp'])"
  done
done 2>&1 | tee gpurun_out/r4_gelu2_ab.log
