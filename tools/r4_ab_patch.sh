#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do
  for pp in 2 1; do
    python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-parity-mode --option fp16_patch_planes=$pp 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('fp16 patch_planes=$pp', d['value'], 'one_stream', d['one_stream']['value'], 'patch', k['patch_embed'], 'parity', d['parity']['argmax_flips'], d['parity']['max_abs_dlogp'])"
  done
done 2>&1 | tee gpurun_out/r4_ab_patch.log
python -m pytest tests/test_fp16_gpu.py -m gpu -q -k "g3_vits8_480_fp16_mode" -s 2>&1 | grep -E "^fp16|passed|failed" | head -12
