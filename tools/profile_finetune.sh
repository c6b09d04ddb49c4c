#!/bin/bash
# rocprofv3 kernel stats of the fine-tune bench (GPU box): bash tools/profile_finetune.sh <tag> [precision]
set -o pipefail
TAG=${1:-ft}; PREC=${2:-bf16x3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ft -- python3 $ROOT/bench.py --mode finetune --precision $PREC --steps 5 --warmup 2 > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
tail -1 $OUT/run.log | cut -c1-200
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/trace/ft_kernel_stats.csv")))
for r in rows[:22]:
    print(f'{r["Name"][:95]:95s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):6.2f} %')
PY
rm -f $OUT/trace/ft_kernel_trace.csv
