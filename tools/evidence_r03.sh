#!/bin/bash
# bash tools/evidence_r03.sh   (GPU box)  one JSON line per BASELINE.json config + rocprofv3 kernel-stat summaries -> gpurun_out/ev_r03/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ev_r03
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_headline.json 2>$OUT/bench_headline.err || tail -5 $OUT/bench_headline.err
python3 bench.py --config parity --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_parity.json 2>/dev/null
python3 bench.py --config 960 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode > $OUT/bench_960.json 2>/dev/null
python3 bench.py --config vitb --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode > $OUT/bench_vitb.json 2>/dev/null
python3 bench.py --config finetune --steps 10 --warmup 3 > $OUT/bench_finetune_bf16.json 2>/dev/null
python3 bench.py --config finetune --precision bf16x3 --steps 6 --warmup 2 > $OUT/bench_finetune_bf16x3.json 2>/dev/null
python3 bench.py --blocks 3 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > $OUT/bench_L3.json 2>/dev/null
python3 bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > $OUT/bench_one_stream.json 2>/dev/null
python3 bench.py --option mlp_fused=0 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > $OUT/bench_unfused_mlp.json 2>/dev/null
python3 bench.py --option proj_fused=0 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode > $OUT/bench_unfused_proj.json 2>/dev/null
python3 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-parity-mode > $OUT/bench_rehearsal2.json 2>$OUT/bench_rehearsal2.err
python3 bench.py --gpus 2 --config finetune --steps 4 --warmup 2 > $OUT/bench_rehearsal2_finetune.json 2>$OUT/bench_rehearsal2_finetune.err
for f in $OUT/bench_*.json; do echo "== $f"; tail -1 $f | cut -c1-260; done
export TMPDIR=/tmp
cd /tmp
for cfg in "960:--config 960 --steps 3 --warmup 1" "vitb:--config vitb --steps 4 --warmup 1" "finetune:--config finetune --steps 4 --warmup 2" "parity:--config parity --steps 3 --warmup 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o b -- python3 $ROOT/bench.py $args --no-cpu-baseline --no-parity-mode --streams 1 > $OUT/trace_$name.log 2>&1 || tail -3 $OUT/trace_$name.log
  find $OUT/trace_$name -name "*kernel_trace.csv" -delete
done
du -sh $OUT
