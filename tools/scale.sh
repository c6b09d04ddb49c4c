#!/bin/bash
# bash tools/scale.sh [--dry-run] [OUT.jsonl]      (an 8-GPU MI355X node; one command from lease to curve)
# The 1 -> 8 GPU scaling curve of both multi-GPU configurations of BASELINE.json, one JSON line per run, appended to OUT
# (default gpurun_out/scale_r04.jsonl):
#   headline  : ViT-S/8 x12 @480, batch 32 per GPU, N independent replicas (no data-path collective)            -- N = 1, 2, 4, 8
#   finetune  : 3-block unfrozen step, batch 8 per GPU, RCCL gradient reduction over xGMI, both collective forms -- N = 1, 2, 4, 8
# Every N > 1 run is `python bench.py --gpus N ...`, which starts its own ranks (a torch.distributed.run child on 127.0.0.1, one rank
# per GPU, backend nccl = RCCL).  --dry-run exercises the launcher, the rendezvous and the JSON plumbing only (gloo, no model, no
# GPU): tests/test_parallel_cpu.py runs it on the CPU.  Nothing here is run by the builder on real GPUs: the pool gives one GPU per
# box; the file exists so that the first 8-GPU lease yields the curve without new code.
set -o pipefail
cd "$(dirname "$0")/.."
DRY=""
if [ "$1" = "--dry-run" ]; then DRY="--dry-run"; shift; fi
OUT=${1:-gpurun_out/scale_r04.jsonl}
GPUS=${SCALE_GPUS:-"1 2 4 8"}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
run() {      # run <label> <bench.py args...>: the run's one JSON line, tagged with the label, appended to OUT
  local label=$1; shift
  local line
  line=$(python bench.py "$@" $DRY --no-cpu-baseline 2>>"$OUT.stderr" | grep '^{' | tail -1)
  if [ -z "$line" ]; then echo "scale.sh: $label produced no JSON line (see $OUT.stderr)" >&2; fail=1; return; fi
  python -c "import json,sys; d=json.loads(sys.argv[1]); d['scale_label']=sys.argv[2]; print(json.dumps(d))" "$line" "$label" >> "$OUT"
  echo "$label: $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], d['unit'], 'ranks_seen', d.get('ranks_seen'))" "$line")"
}
for n in $GPUS; do
  run "headline N=$n" --gpus $n --steps 20 --warmup 5 --no-parity-mode
done
for coll in allreduce rs_ag; do
  for n in $GPUS; do
    run "finetune $coll N=$n" --gpus $n --config finetune --collective $coll --steps 20 --warmup 5
  done
done
# efficiency is the reader's to compute (value(N) / (N * value(1))); this only lists the points
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
print(f"{len(rows)} runs -> {sys.argv[1]}")
for r in rows:
    print(f"  {r['scale_label']:28s} n_gpus {r['n_gpus']}  {r['value']:10.2f} {r['unit']}  ranks_seen {r.get('ranks_seen')}")
PY
exit $fail
