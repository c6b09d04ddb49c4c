#!/bin/bash
# GPU box: timing ablations of gemm_rs.hip (variants built with tools/build_variant1.sh rsabl<N> gemm_rs -DRS_ABL=<N>)
for v in "" ${RS_VARIANTS:-rsabl2 rsabl16 rsabl1 rsabl32 rsabl18}; do
  if [ -z "$v" ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$PWD/dino_amd/lib/variants/lib_$v.so; fi
  echo "== ${v:-tree}"; timeout -k 10 120 python tools/bench_rs.py 57616 10 2>&1 | grep -E "^(qkv|fc1|proj|fc2)" | awk 'NR%2==0'
done
