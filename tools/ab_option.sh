#!/bin/bash
# bash tools/ab_option.sh <rounds> <label=opt1=v1[,opt2=v2]> ...   (GPU box)  interleaved `python bench.py` runs of the in-tree build under
# different library options on ONE box (boxes differ by +-5 %); "base" = no option.  Prints frames/s, the attention launch and the step's classes.
#   bash tools/ab_option.sh 3 base old=attn_variant=11 mq1=attn_variant=1035
R=$1; shift
EXTRA=${AB_EXTRA:---steps 15 --warmup 4 --no-cpu-baseline --no-parity-mode --no-configs}
for i in $(seq 1 $R); do
  for spec in "$@"; do
    label=${spec%%=*}; opts=""
    if [ "$spec" != "$label" ]; then for kv in $(echo ${spec#*=} | tr ',' ' '); do opts="$opts --option $kv"; done; fi
    python bench.py $EXTRA $opts 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step',{})
print('$label', d['value'], 'one_stream', (d.get('one_stream') or {}).get('value'), 'attn_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'qkv', k.get('qkv_gemm'), 'mlp', k.get('fc1_gemm'))"
  done
done
