#!/bin/bash
# fused MLP: op parity, then the headline with and without it, one and two streams (same box)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "mlp_fused" 2>&1 | tail -2
for r in 1 2; do
for cfg in "fused1:--streams 1" "unfused1:--streams 1 --option mlp_fused=0" "fused2:--streams 2" "unfused2:--streams 2 --option mlp_fused=0"; do
  n=${cfg%%:*}; a=${cfg#*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode --no-two-stream --profile-all $a > gpurun_out/r3_bench_$n.json 2> gpurun_out/r3_bench_$n.err || echo "bench $n failed"
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/r3_bench_{n}.json").read().strip().splitlines()[-1])
    k = j.get("kernel_ms_per_step")
    print(n, j["value"], j["ms_per_step"], {a: k[a] for a in ("qkv_gemm", "attention", "proj_gemm", "fc1_gemm", "fc2_gemm")}, j.get("parity", {}).get("argmax_flips"))
except Exception as e:
    print(n, "ERR", e)
PY
done
done
