#!/bin/bash
# round 4, GPU call 4: small-batch dispatch (LayerNorm-fused 128x384 panels leave most CUs idle below ~100 panels?)
set -o pipefail
mkdir -p gpurun_out
for B in 1 2 3 4 6; do
  for gl in 0 1; do
    for prec in bf16 bf16x3; do
    python bench.py --batch $B --precision $prec --streams 1 --option gemm_ln=$gl --steps 40 --warmup 5 --no-cpu-baseline --no-parity-mode --no-two-stream 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('B=$B gemm_ln=$gl $prec', d['value'], 'ms/step', d['ms_per_step'], {a: round(b,3) for a,b in k.items()})"
    done
  done
done 2>&1 | tee gpurun_out/r4_smallbatch.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('headline', d['value'], 'one_stream', d['one_stream']['value'], k)"
