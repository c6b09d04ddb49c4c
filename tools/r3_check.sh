#!/bin/bash
# full -m gpu suite, the default bench line, the MFMA shape measurement
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_gputests.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r3_bench_default.json").read().strip().splitlines()[-1])
    print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["kernel_ms_per_step"], j.get("one_stream"), j["parity"], j.get("parity_mode", {}).get("value"), j.get("parity_mode", {}).get("parity"), j.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("ERR", e)
PY
MFMA_MODES=500,504,604,704 timeout -k 10 300 python tools/mfma_peak.py > gpurun_out/r3_mfma_shapes.txt 2>&1; echo "mfma rc=$?"; cat gpurun_out/r3_mfma_shapes.txt
