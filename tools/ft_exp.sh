#!/bin/bash
# bash tools/ft_exp.sh <tag> [trace] [-- option sets...]   (GPU box)
# Fine-tune step (configs[3] per-GPU shape: 3 blocks, 8 frames @480): one bench line per option set ("k=v,k=v" or "-"), then, with
# "trace" as the second argument, a rocprofv3 --kernel-trace --stats pass of the default build -> gpurun_out/ft_<tag>/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
TRACE=0
if [ "$1" = "trace" ]; then TRACE=1; shift; fi
OUT=$ROOT/gpurun_out/ft_$TAG
mkdir -p $OUT
cd $ROOT
[ $# -eq 0 ] && set -- -
for set_ in "$@"; do
  opts=""
  if [ "$set_" != "-" ]; then for kv in ${set_//,/ }; do opts="$opts --option $kv"; done; fi
  python3 bench.py --config finetune --steps 20 --warmup 4 $opts > $OUT/line_${set_//[,=]/_}.json 2>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  echo "$set_ : $(python3 -c "import json,sys; d=json.loads(open('$OUT/line_${set_//[,=]/_}.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
if [ $TRACE = 1 ]; then
  export TMPDIR=/tmp
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- python3 $ROOT/bench.py --config finetune --steps 6 --warmup 2 > $OUT/trace.log 2>&1 || tail -3 $OUT/trace.log
  find $OUT/trace -name "*kernel_trace.csv" -delete
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/b_kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("per step (8 steps traced): total %.3f ms" % (tot/8e6))
for r in rows[:28]:
    print("%-60s %4d %8.1f us/call %8.1f us/step" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/8e3))
PY
fi
