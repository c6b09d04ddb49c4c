#!/bin/bash
# bash tools/ab_ops.sh <bench_ops mode> <rounds> <libA.so> <libB.so> ...   (GPU box)  the same micro-benchmark through several
# builds of the library, interleaved on one box ("cur" = the in-tree build)
MODE=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = cur ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$lib; fi
    python tools/bench_ops.py $MODE 2>/dev/null | sed "s#^#$(basename $lib .so) | #"
  done
done
