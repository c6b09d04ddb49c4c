#!/bin/bash
# fused-MLP ablation variants (tools/build_variant.sh mfablN "-DMF_ABL=N"), interleaved on one box:  r3_abl.sh <rows> <variant> ...
mkdir -p gpurun_out
M=$1; shift
for r in 1 2; do
  timeout -k 10 120 python tools/bench_mlp.py $M 30 2>&1 | tail -1
  for v in "$@"; do
    DINOSEG_LIB=build/variants/lib_mfabl$v.so timeout -k 10 120 python tools/bench_mlp.py $M 30 2>&1 | tail -1
  done
done
