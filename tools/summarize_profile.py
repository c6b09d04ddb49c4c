#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh on the GPU box) into profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of the bench command (dseg kernels + top others)
  profiles/<tag>_pmc_hbm.csv        per-kernel mean FETCH_SIZE / WRITE_SIZE from the two separate --pmc passes
  profiles/<tag>_summary.md         the numbers the bench line quotes, with the gfx950 corrections spelled out
  profiles/attention_traffic.json   HBM bytes per attention launch (read by bench.py for roofline.traffic)

HBM accounting follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read, so the read side is
doubled; WRITE_SIZE reads exactly for 16-B-per-lane stores.  Narrower accesses are uncalibrated -- the figure
is an estimate to compare against the algorithmic bytes, not an absolute.
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0] if "dseg::" in name else name[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    old_traffic = {}
    if os.path.exists(os.path.join(dst, "attention_traffic.json")):
        old_traffic = json.load(open(os.path.join(dst, "attention_traffic.json")))

    rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
    keep = [r for r in rows if "dseg::" in r["Name"]] + [r for r in rows if "dseg::" not in r["Name"]][:3]
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in keep:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"],
                        r["MinNs"], r["MaxNs"]])

    pmc = defaultdict(lambda: defaultdict(list))
    for which in ("fetch", "write"):
        path = os.path.join(src, f"pmc_{which}", "bench_counter_collection.csv")
        if not os.path.exists(path):
            continue
        for r in csv.DictReader(open(path)):
            if "dseg::" in r["Kernel_Name"]:
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    hbm = {}
    with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "launches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "read_bytes_corrected(x2)",
                    "write_bytes", "hbm_bytes_per_launch"])
        for k, d in sorted(pmc.items()):
            fe = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [])), 1)
            wr = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [])), 1)
            rb, wb = fe * 1024 * 2, wr * 1024
            hbm[k] = rb + wb
            w.writerow([k, len(d.get("FETCH_SIZE", [])), f"{fe:.1f}", f"{wr:.1f}", f"{rb:.0f}", f"{wb:.0f}", f"{rb + wb:.0f}"])

    bench_line = None
    log = os.path.join(src, "trace_bench.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{") and '"metric"' in line:
                bench_line = json.loads(line)
    # the bf16 (one plane) forward attention kernel of the timed loop: attention_z.hip since round 2, attention.hip before
    # (the instantiation with the most launches: the batch's; the single-frame fixture check launches the narrow one a few times)
    launches = {k: len(d.get("FETCH_SIZE", [])) for k, d in pmc.items()}
    # (round 5: attention_za.hip's assembly tile loop from four rounds of workgroups on)
    cands = ([k for k in hbm if "attn_fwd_za_kernel<" in k and "true" not in k] or [k for k in hbm if "attn_fwd_z_kernel<1" in k]
             or [k for k in hbm if "attn_fwd_kernel<1" in k])
    att = max(cands, key=lambda k: launches.get(k, 0)) if cands else None
    if att and bench_line:
        cfg = bench_line["config"]
        json.dump({"tag": tag, "kernel": att, "hbm_bytes_per_launch": hbm[att], "batch": cfg["batch_per_gpu"],
                   "resolution": cfg["resolution"], "precision": cfg["precision"],
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = 2*FETCH_SIZE*1024 + "
                             "WRITE_SIZE*1024 (gfx950 FETCH_SIZE half-count correction)"},
                  open(os.path.join(dst, "attention_traffic.json"), "w"), indent=1)
        # (the clock fields are measured by a different pass -- tools/pmc_ops.sh -- and carried over)

    if old_traffic:
        tj = json.load(open(os.path.join(dst, "attention_traffic.json")))
        for key, val in old_traffic.items():
            if key.startswith("clock_") or key.startswith("measured_"):
                tj[key] = val
        # the clock of THIS round's kernel when its pass is there (tools/evidence.sh -> profiles/<tag>_clock_kernels.txt)
        ck = os.path.join(dst, f"{tag}_clock_kernels.txt")
        if att and os.path.exists(ck):
            for line in open(ck):
                if line.startswith(att + " ") and "clock_GHz=" in line:
                    tj["clock_ghz_under_load"] = float(line.rsplit("clock_GHz=", 1)[1])
                    tj["clock_method"] = (f"rocprofv3 --pmc GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration of the same pass "
                                          f"(tools/pmc_cmd.sh clk_{tag} ... bench.py --streams 1, profiles/{tag}_clock_kernels.txt): "
                                          + line.strip()[:400])
        json.dump(tj, open(os.path.join(dst, "attention_traffic.json"), "w"), indent=1)

    with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 "
                f"--no-cpu-baseline --no-parity-mode --no-configs --streams 1` (tools/profile_bench.sh), MI355X, one GPU.  (`--streams 1`: the default "
                f"line times two half-batches on two streams, whose launches overlap; here every launch has the chip to itself, like "
                f"the untimed roofline pass of the default line, so that the per-kernel averages below are those the roofline leg quotes.)\n\n")
        if bench_line:
            f.write("Bench line under the profiler (profiled clocks run ~2-3 % lower than un-profiled):\n\n```\n"
                    + json.dumps(bench_line) + "\n```\n\n")
        f.write("| kernel | calls | avg us | % of GPU time |\n|---|---|---|---|\n")
        for r in keep:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        f.write("\nHBM traffic per launch (PMC, separate passes; read side doubled per the gfx950 FETCH_SIZE correction):\n\n"
                "| kernel | HBM MB / launch |\n|---|---|\n")
        for k, v in sorted(hbm.items(), key=lambda kv: -kv[1]):
            f.write(f"| `{k}` | {v / 1e6:.1f} |\n")
    print(open(os.path.join(dst, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
