#!/bin/bash
# GPU box: parity mode (fp16x3) frames/s against the batch size, the hi + lo fused launch off / on at every size (option mlp_fused 0 / 2),
# one stream and the library default -- where does mlp_fused_min_rows belong?
for B in 1 2 3 4 6 8 12 16; do
  for mf in 0 2; do
    timeout -k 10 200 python bench.py --config parity --batch $B --steps 10 --warmup 3 --no-cpu-baseline --option mlp_fused=$mf 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$B mlp_fused=$mf', d['value'], 'one stream', (d.get('one_stream') or {}).get('value'), 'dlogp', d['parity']['max_abs_dlogp'], 'flips', d['parity']['argmax_flips'])"
  done
done
