#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash tools/profile_bench.sh <tag> [bench args...]
# 1. rocprofv3 --kernel-trace --stats of the bench command on ONE stream (--streams 1: exclusive launches, the shape of the
#    roofline leg; the default line times two half-batches on two streams) -> gpurun_out/prof_<tag>/
# 2. separate PMC passes (FETCH_SIZE, WRITE_SIZE) -- never combined with tracing domains other than kernel-trace
# Summaries worth judging are copied into profiles/ by tools/summarize_profile.py afterwards (in the build container).
set -o pipefail
TAG=${1:-r01}
shift
ARGS=${@:---steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-configs --streams 1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ROOT/bench.py $ARGS > $OUT/trace_bench.log 2>&1 || { tail -20 $OUT/trace_bench.log; exit 1; }
tail -1 $OUT/trace_bench.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-configs --streams 1 > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --no-configs --streams 1 > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
# keep only what fits the 64 MiB merge budget
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
