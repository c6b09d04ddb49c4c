#!/bin/bash
# bash tools/evidence.sh <tag>   (GPU box)  one JSON line per BASELINE.json config + rocprofv3 kernel-stat summaries + PMC passes
# -> gpurun_out/ev_$TAG/ and gpurun_out/prof_$TAG/ ; condensed into profiles/$TAG_* by tools/collect_evidence.py <tag> and
# tools/summarize_profile.py <tag> (in the build container)
set -o pipefail
TAG=${1:?usage: bash tools/evidence.sh <tag, e.g. r05>}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ev_$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_headline.json 2>$OUT/bench_headline.err || tail -5 $OUT/bench_headline.err     # (the driver's command: configs sub-records included)
python3 bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > $OUT/bench_headline_bf16.json 2>/dev/null
python3 bench.py --config parity --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_parity.json 2>/dev/null
python3 bench.py --config parity --precision bf16x3 --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_parity_bf16x3.json 2>/dev/null
python3 bench.py --config 960 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode > $OUT/bench_960.json 2>/dev/null
python3 bench.py --config vitb --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode > $OUT/bench_vitb.json 2>/dev/null
python3 bench.py --config finetune --steps 10 --warmup 3 > $OUT/bench_finetune_bf16.json 2>/dev/null
python3 bench.py --config finetune --precision bf16x3 --steps 6 --warmup 2 > $OUT/bench_finetune_bf16x3.json 2>/dev/null
python3 bench.py --blocks 3 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > $OUT/bench_L3.json 2>/dev/null
python3 bench.py --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-parity-mode --no-two-stream > $OUT/bench_B1.json 2>/dev/null
DINOSEG_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode > $OUT/bench_rehearsal2.json 2>$OUT/bench_rehearsal2.err
DINOSEG_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --config finetune --steps 4 --warmup 2 > $OUT/bench_rehearsal2_finetune.json 2>$OUT/bench_rehearsal2_finetune.err
for f in $OUT/bench_*.json; do echo "== $f"; tail -1 $f | cut -c1-200; done
python3 tools/latency_b1.py bf16x3,fp16x3,fp16 > $OUT/latency_b1.txt 2>/dev/null; grep "^L=" $OUT/latency_b1.txt
export TMPDIR=/tmp
cd /tmp
for cfg in "960:--config 960 --steps 3 --warmup 1" "vitb:--config vitb --steps 4 --warmup 1" "finetune:--config finetune --steps 4 --warmup 2" "parity:--config parity --steps 3 --warmup 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o b -- python3 $ROOT/bench.py $args --no-cpu-baseline --no-parity-mode --no-configs --streams 1 > $OUT/trace_$name.log 2>&1 || tail -3 $OUT/trace_$name.log
  find $OUT/trace_$name -name "*kernel_trace.csv" -delete
done
cd $ROOT
bash tools/profile_bench.sh $TAG | tail -3
bash tools/pmc_cmd.sh clk_$TAG "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-mode --no-configs --streams 1 > $OUT/clock_kernels.txt 2>&1; grep "dseg::" $OUT/clock_kernels.txt | cut -c1-220
du -sh $OUT $ROOT/gpurun_out/prof_$TAG
