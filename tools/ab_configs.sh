#!/bin/bash
# GPU box: `bench.py --config <c>` of the working tree's library against dino_amd/lib/variants/lib_<name>.so, interleaved on one box.
#   bash tools/ab_configs.sh <name> <rounds> <config> [<config> ...]        e.g.  bash tools/ab_configs.sh pre 2 parity vitb
NAME=$1; R=$2; shift 2
BASE=${GRAFT_REPO_ROOT:-$(pwd)}/dino_amd/lib/variants/lib_$NAME.so
for r in $(seq $R); do
  for c in "$@"; do
    for lib in "" "$BASE"; do
      if [ -z "$lib" ]; then unset DINOSEG_LIB; tag=tree; else export DINOSEG_LIB=$lib; tag=$NAME; fi
      timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-configs --no-parity-mode 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '$c', d['value'], d['ms_per_step'], (d.get('parity') or d.get('gradient_parity') or {}).get('max_abs_dlogp'))"
    done
  done
done
