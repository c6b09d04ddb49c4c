#!/bin/bash
# bash tools/build_variant.sh <name> "<extra hipcc flags>"  ->  dino_amd/lib/variants/lib_<name>.so   (A/B builds of the HIP library:
# git-ignored like every .so, but NOT gpurun-ignored -- build/ is -- so they travel to the GPU box;
# run side by side on one GPU box with DINOSEG_LIB=... : boxes differ by +-5 %)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/build/variant_$NAME
mkdir -p $B $ROOT/dino_amd/lib/variants
for f in api train_api gemm gemm_big gemm_ln gemm_ln12 mlp_fused2 gemm_tn attention attention_z attention_za attention_bwd elementwise train; do
  SLP=""; case $f in mlp_fused*) SLP="-fno-slp-vectorize";; esac      # (as in the Makefile)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $SLP $EXTRA -c $ROOT/dino_amd/csrc/$f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/dino_amd/lib/variants/lib_$NAME.so $B/*.o
echo built $ROOT/dino_amd/lib/variants/lib_$NAME.so
