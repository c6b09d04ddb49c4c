#!/bin/bash
# same-box comparison of the round-3 tree (build/r03_tree, bf16 headline) with this tree (fp16 default, bf16), interleaved
set -o pipefail
mkdir -p gpurun_out
one() {  # one <label> <dir> <args...>
  local label=$1 dir=$2; shift 2
  (cd $dir && python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-parity-mode "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$label', d['value'], 'one_stream', d['one_stream']['value'], 'patch', k['patch_embed'], 'qkv', k['qkv_gemm'], 'attn', k['attention'], 'mlp', k['fc1_gemm'], 'head', k['head'])")
}
for i in 1 2 3; do
  one r03_bf16 build/r03_tree
  one r04_bf16 . --precision bf16
  one r04_fp16 .
done 2>&1 | tee gpurun_out/r4_ab_r03.log
