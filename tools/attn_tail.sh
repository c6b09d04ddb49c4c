#!/bin/bash
# GPU box: the attention launch against the batch size -- is the time a staircase in rounds of workgroups (512 slots: 256 CUs x 2) or
# linear in the work?  90 workgroups per frame @480 (6 heads x 15 q-tiles of 256 queries).  bash tools/attn_tail.sh [variants]
V=${1:-66571}
for B in 11 17 22 23 28 29 32 34 35 40; do
  ATTN_B=$B ATTN_PLANES=1 ATTN_VARIANTS=$V python tools/bench_ops.py attn 2>/dev/null | grep "attention B="
done
