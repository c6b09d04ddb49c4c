#!/bin/bash
# GPU box: interleaved A/B of the hi + lo fused launch alone and of the parity line: in-tree build vs variants / options
#   bash tools/m3_ab.sh "<variant names>" "<option strings, e.g. mlp_grid=256>"
for r in 1 2; do
  for v in "" $1; do
    if [ -z "$v" ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$PWD/dino_amd/lib/variants/lib_$v.so; fi
    timeout -k 10 120 python tools/bench_mlp3.py 115232 20 1 1 2>&1 | grep mlp_fused3 | tail -1
  done
done
unset DINOSEG_LIB
for r in 1 2; do
  for o in "" $2; do
    opt=""; [ -n "$o" ] && opt="--option $o"
    timeout -k 10 300 python bench.py --config parity --no-cpu-baseline --steps 10 --warmup 3 $opt 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('parity [$o]', d['value'], 'one stream', d['one_stream']['value'], 'fused ms', d['kernel_ms_per_step']['fc1_gemm'])"
  done
done
