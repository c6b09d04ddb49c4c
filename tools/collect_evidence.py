#!/usr/bin/env python3
"""python tools/collect_evidence.py [tag, default r04]
gpurun_out/ev_<tag>/ (tools/evidence_<tag>.sh on the GPU box) -> profiles/<tag>_bench_lines.jsonl + profiles/<tag>_<config>_kernel_stats.csv
(+ clock_kernels.txt and latency_b1.txt when present)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", f"ev_{TAG}")
DST = os.path.join(ROOT, "profiles")
lines = []
for name in ("headline", "headline_bf16", "parity", "parity_bf16x3", "960", "vitb", "finetune_bf16", "finetune_bf16x3", "L3", "B1", "one_stream", "unfused_mlp",
             "unfused_proj", "rehearsal2", "rehearsal2_finetune"):
    p = os.path.join(SRC, f"bench_{name}.json")
    if os.path.exists(p):
        txt = [l for l in open(p).read().strip().splitlines() if l.startswith("{")]
        if txt:
            d = json.loads(txt[-1])
            d["evidence_config"] = name
            lines.append(json.dumps(d))
open(os.path.join(DST, f"{TAG}_bench_lines.jsonl"), "w").write("\n".join(lines) + "\n")
for name in ("960", "vitb", "finetune", "parity"):
    p = os.path.join(SRC, f"trace_{name}", "b_kernel_stats.csv")
    if not os.path.exists(p):
        continue
    rows = list(csv.DictReader(open(p)))
    keep = [r for r in rows if "dseg::" in r["Name"]] + [r for r in rows if "dseg::" not in r["Name"]][:3]
    with open(os.path.join(DST, f"{TAG}_{name}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in keep:
            w.writerow([r["Name"].replace("void ", "").split("(")[0] if "dseg::" in r["Name"] else r["Name"][:80], r["Calls"],
                        r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
for extra, dst in (("clock_kernels.txt", f"{TAG}_clock_kernels.txt"), ("latency_b1.txt", f"{TAG}_latency_b1.txt")):
    if os.path.exists(os.path.join(SRC, extra)):
        keep = [l for l in open(os.path.join(SRC, extra)) if l.startswith("dseg::") or l.startswith("L=")]
        open(os.path.join(DST, dst), "w").write("".join(keep))
print(len(lines), "bench lines")
