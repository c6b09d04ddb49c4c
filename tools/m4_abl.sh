#!/bin/bash
# GPU box: the one-wave fused launch (mlp_fused4.hip) alone through ablation / experiment builds made by
#   for a in 1 2 16 32 ...; do bash tools/build_variant1.sh m4abl$a mlp_fused4 -DMF4_ABL=$a; done
#   bash tools/m4_abl.sh "<variant names>" [rows]
M=${2:-115232}
for r in 1 2; do
for v in "" $1; do
  if [ -z "$v" ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$PWD/dino_amd/lib/variants/lib_$v.so; fi
  timeout -k 10 120 python tools/bench_mlp4.py $M 20 1 1 4 2>&1 | grep mlp_fused4 | tail -1
done; done
