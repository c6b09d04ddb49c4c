#!/bin/bash
# GPU box: the hi + lo assembly attention kernel: bit-identity against the compiled zero-reference hi + lo kernel, then the three hi + lo
# kernels interleaved (11 = attention.hip reference-based, 27 = attention_z.hip<2,3,12>, 1051 = attention_za.hip X3), bf16 planes
OUT=${1:-gpurun_out/za3x}
mkdir -p $OUT
timeout -k 10 420 python -m pytest tests/test_ops_gpu.py -k "attention_za" -x -q > $OUT/tests.log 2>&1
rc=$?
tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
ATTN_PLANES=2 ATTN_VARIANTS=11,27,1051 timeout -k 10 300 python tools/bench_ops.py attn > $OUT/bench_x3.log 2>&1 || exit 1
cat $OUT/bench_x3.log
