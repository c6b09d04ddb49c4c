#!/bin/bash
# GPU box: the working tree's library against dino_amd/lib/variants/lib_<name>.so (tools/build_variant.sh in a worktree of the commit to
# compare with), attention launches interleaved:  bash tools/ab_lib.sh <name> [rounds]
NAME=${1:-base}; R=${2:-2}
BASE=$GRAFT_REPO_ROOT/dino_amd/lib/variants/lib_$NAME.so
[ -f "$BASE" ] || BASE=$(pwd)/dino_amd/lib/variants/lib_$NAME.so
for r in $(seq $R); do
  for lib in "" "$BASE"; do
    tag=$([ -z "$lib" ] && echo "tree" || echo "$NAME")
    if [ -z "$lib" ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$lib; fi
    OP_FMT=1 ATTN_PLANES=1 ATTN_VARIANTS=66571 timeout -k 10 200 python tools/bench_ops.py attn 2>&1 | grep attention | sed "s/^/$tag /"
    OP_FMT=0 ATTN_PLANES=2 ATTN_VARIANTS=66571 timeout -k 10 200 python tools/bench_ops.py attn 2>&1 | grep attention | sed "s/^/$tag /"
  done
done
