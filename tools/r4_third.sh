#!/bin/bash
# round 4, GPU call 3: calibrations + bench (fp16 default) + single-frame latency + the non-temporal experiments
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "adam_steps or bf16_train" > gpurun_out/r4_tests2.log 2>&1
grep -E "adam traj|bf16 step|passed|failed" gpurun_out/r4_tests2.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench1.log 2>&1 && tail -1 gpurun_out/r4_bench1.log > gpurun_out/r4_bench1.json && python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench1.json'))
print('value', d['value'], d['dtype'], 'parity', d['parity'])
print('bf16_mode', d.get('bf16_mode'))
print('parity_mode', d.get('parity_mode'))
print('roofline', d['roofline']['achieved'], d['roofline']['avg_launch_ms'], 'kernels', d['kernel_ms_per_step'], 'one_stream', d.get('one_stream'))
PY
python tools/latency_b1.py > gpurun_out/r4_latency.log 2>&1 && cat gpurun_out/r4_latency.log
for i in 1 2; do
  for lib in cur nt1 nt2 nt3; do
    if [ $lib = cur ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=build/variants/lib_$lib.so; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', d['value'], 'one_stream', d['one_stream']['value'], 'qkv', k['qkv_gemm'], 'attn', k['attention'], 'mlp', k['fc1_gemm'])"
  done
done 2>&1 | tee gpurun_out/r4_nt_ab.log
