#!/bin/bash
# GPU box: the one-wave fused launch against the number of items per workgroup (256 workgroups x 128 rows x n): the increment per item is an
# item's compute + its boundary, without the launch's start and tail.   bash tools/m4_items.sh "<variant names>"
for v in "" $1; do
  if [ -z "$v" ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$PWD/dino_amd/lib/variants/lib_$v.so; fi
  for n in 1 2 4 8; do
    timeout -k 10 120 python tools/bench_mlp4.py $((32768 * n)) 20 1 1 ${WHICH:-4} 2>&1 | grep mlp_fused | tail -${TAILN:-1}
  done
done
