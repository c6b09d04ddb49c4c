#!/bin/bash
# bash tools/ab_bench.sh <old.so> [rounds]   (GPU box)  interleaved bench.py runs of an older build (DINOSEG_LIB) and the
# in-tree build on the SAME box: boxes differ by +-5 %, so a number from one box says nothing about a change.
OLD=$1; R=${2:-3}
for i in $(seq 1 $R); do
  for which in old new; do
    if [ $which = old ]; then export DINOSEG_LIB=$OLD; else unset DINOSEG_LIB; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', d['value'], 'attn_ms', d['roofline']['avg_launch_ms'])"
  done
done
