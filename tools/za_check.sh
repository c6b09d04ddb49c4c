#!/bin/bash
# GPU box: bit-identity of the assembly attention kernel against the compiled one, then both interleaved in one process.
#   bash tools/za_check.sh [out_dir]
OUT=${1:-gpurun_out/za}
mkdir -p $OUT
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -k "attention_za or pos_embed or another_layout" -x -q > $OUT/tests.log 2>&1
rc=$?
tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
for fmt in 0 1; do
  OP_FMT=$fmt ATTN_PLANES=1 ATTN_VARIANTS=${ZA_VARIANTS:-11,1035,66571} timeout -k 10 300 python tools/bench_ops.py attn > $OUT/bench_fmt$fmt.log 2>&1 || exit 1
  cat $OUT/bench_fmt$fmt.log
done
