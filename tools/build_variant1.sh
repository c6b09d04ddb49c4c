#!/bin/bash
# bash tools/build_variant1.sh <name> <file (without .hip)> "<extra hipcc flags>"  ->  dino_amd/lib/variants/lib_<name>.so
# A/B build of ONE translation unit (ablation / experiment macros), linked with the tree's other objects (run `make` first).
set -e
NAME=$1; F=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/build/variant1_$NAME
mkdir -p $B $ROOT/dino_amd/lib/variants
SLP=""; case $F in mlp_fused*) SLP="-fno-slp-vectorize";; esac      # (as in the Makefile)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $SLP $EXTRA"
/opt/rocm/bin/hipcc $FLAGS -c $ROOT/dino_amd/csrc/$F.hip -o $B/$F.o &
if [ "$F" = mlp_fused3 ]; then /opt/rocm/bin/hipcc $FLAGS -DMF3_PART=1 -c $ROOT/dino_amd/csrc/$F.hip -o $B/${F}_b.o & fi      # (its second object: Makefile)
wait
OBJS=$(ls $ROOT/build/csrc/*.o | grep -v "/$F.o" | grep -v "/${F}_b.o" | grep -v -- "-hip-amdgcn" | grep -v -- "-host-")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/dino_amd/lib/variants/lib_$NAME.so $B/*.o $OBJS
echo built $ROOT/dino_amd/lib/variants/lib_$NAME.so
