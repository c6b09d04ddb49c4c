#!/bin/bash
# fused-MLP ablation variants (tools/build_variant.sh <name> "-DMF_ABL=N" / "-DMF2_ABL=N"), interleaved on one box:
#   r3_abl.sh <rows> <kernel variant 1|2> <lib name> ...
mkdir -p gpurun_out
M=$1; V=$2; shift 2
for r in 1 2; do
  timeout -k 10 120 python tools/bench_mlp.py $M 30 $V 2>&1 | tail -1
  for v in "$@"; do
    DINOSEG_LIB=dino_amd/lib/variants/lib_$v.so timeout -k 10 120 python tools/bench_mlp.py $M 30 $V 2>&1 | tail -1
  done
done
