#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for prec in fp16 fp16x3 bf16x3 bf16; do
  timeout -k 10 300 python tools/fuzz_routes.py 16 4 $prec > gpurun_out/r4_fuzz_$prec.log 2>&1; rc=$?
  echo "== $prec rc=$rc"; tail -3 gpurun_out/r4_fuzz_$prec.log | cut -c1-300
  [ $rc -eq 0 ] || exit 1
done
timeout -k 10 300 python tools/race_screen.py 20 > gpurun_out/r4_race.log 2>&1; echo "race rc=$?"; tail -4 gpurun_out/r4_race.log
timeout -k 10 200 python tools/fuzz_train.py > gpurun_out/r4_fuzz_train.log 2>&1; echo "fuzz_train rc=$?"; tail -3 gpurun_out/r4_fuzz_train.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()"
