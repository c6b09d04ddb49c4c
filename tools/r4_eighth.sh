#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/ab_ops.sh attn 2 cur build/variants/lib_prio1.so build/variants/lib_prio3.so 2>&1 | grep "planes=1" | tee gpurun_out/r4_prio_ab.log
for i in 1 2; do
  for lib in cur prio1; do
    if [ $lib = cur ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=build/variants/lib_$lib.so; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', d['value'], 'one_stream', d['one_stream']['value'], 'attn', k['attention'])"
  done
done 2>&1 | tee -a gpurun_out/r4_prio_ab.log
