#!/bin/bash
# GPU box: which reads are the fused proj+MLP launch's extra FETCH bytes?  FETCH_SIZE / WRITE_SIZE per launch of
# mlp_fused2_kernel<true,...> as built, with the weight stream ablated (-DMF2_ABL=2) and with the row loads / stores ablated (-DMF2_ABL=32);
# variants: bash tools/build_variant.sh mf2abl2 "-DMF2_ABL=2"; bash tools/build_variant.sh mf2abl32 "-DMF2_ABL=32"
for v in base mf2abl2 mf2abl32; do
  if [ $v = base ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$PWD/dino_amd/lib/variants/lib_$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    echo "== $v / $c"
    bash tools/pmc_cmd.sh mlpattr_${v} "$c" tools/bench_mlp.py 115232 6 3 2>&1 | grep "dseg::mlp_fused2"
  done
done
