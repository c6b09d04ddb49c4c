#!/bin/bash
# bash tools/pmc_cmd.sh <tag> "<COUNTER ...>" <python script> [args...]   (GPU box; counters in their own pass, kernel-trace only)
# per-kernel averages of the counters + duration + effective clock for every dseg:: kernel of the command
set -o pipefail
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SCRIPT=$ROOT/$1; shift
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o ops -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
grep -v amdgpu $OUT/run.log | tail -3
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/ops_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "dseg::" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
try:
    for r in csv.DictReader(open("$OUT/ops_kernel_trace.csv")):
        if "dseg::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
except Exception as e:
    print("no kernel trace:", e)
for k, d in agg.items():
    line = {c: round(sum(v) / len(v)) for c, v in d.items()}
    extra = ""
    if k in dur:
        ns = sum(dur[k]) / len(dur[k])
        extra = " avg_us=%.1f" % (ns / 1e3)
        if "GRBM_GUI_ACTIVE" in line:
            extra += " clock_GHz=%.3f" % (line["GRBM_GUI_ACTIVE"] / 8 / ns)
    print(k, line, "n=%d" % len(next(iter(d.values()))), extra)
PY
rm -f $OUT/ops_kernel_trace.csv $OUT/ops_counter_collection.csv
