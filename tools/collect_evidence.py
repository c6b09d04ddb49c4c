#!/usr/bin/env python3
"""gpurun_out/ev_r03/ (tools/evidence_r03.sh on the GPU box) -> profiles/r03_bench_lines.jsonl + profiles/r03_<config>_kernel_stats.csv"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "ev_r03")
DST = os.path.join(ROOT, "profiles")
lines = []
for name in ("headline", "parity", "960", "vitb", "finetune_bf16", "finetune_bf16x3", "L3", "one_stream", "unfused_mlp", "unfused_proj", "rehearsal2", "rehearsal2_finetune"):
    p = os.path.join(SRC, f"bench_{name}.json")
    if os.path.exists(p):
        txt = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
        if txt:
            d = json.loads(txt[-1])
            d["evidence_config"] = name
            lines.append(json.dumps(d))
open(os.path.join(DST, "r03_bench_lines.jsonl"), "w").write("\n".join(lines) + "\n")
for name in ("960", "vitb", "finetune", "parity"):
    p = os.path.join(SRC, f"trace_{name}", "b_kernel_stats.csv")
    if not os.path.exists(p):
        continue
    rows = list(csv.DictReader(open(p)))
    keep = [r for r in rows if "dseg::" in r["Name"]] + [r for r in rows if "dseg::" not in r["Name"]][:3]
    with open(os.path.join(DST, f"r03_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in keep:
            w.writerow([r["Name"].replace("void ", "").split("(")[0] if "dseg::" in r["Name"] else r["Name"][:80], r["Calls"],
                        r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
print(len(lines), "bench lines")
