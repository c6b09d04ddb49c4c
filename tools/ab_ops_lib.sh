#!/bin/bash
# GPU box: `tools/bench_ops.py <what>` of the working tree's library against dino_amd/lib/variants/lib_<name>.so, interleaved.
#   bash tools/ab_ops_lib.sh <name> <rounds> <what> [grep pattern]
NAME=$1; R=$2; WHAT=$3; PAT=${4:-.}
BASE=${GRAFT_REPO_ROOT:-$(pwd)}/dino_amd/lib/variants/lib_$NAME.so
for r in $(seq $R); do
  for lib in "" "$BASE"; do
    if [ -z "$lib" ]; then unset DINOSEG_LIB; tag=tree; else export DINOSEG_LIB=$lib; tag=$NAME; fi
    timeout -k 10 300 python tools/bench_ops.py $WHAT 2>&1 | grep -E "$PAT" | sed "s/^/$tag /"
  done
done
