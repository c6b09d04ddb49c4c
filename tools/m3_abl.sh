for M in 32768 115232; do
for v in "" m3abl1 m3abl2 m3abl16 m3abl32 m3abl3 m3abl19; do
  if [ -z "$v" ]; then unset DINOSEG_LIB; else export DINOSEG_LIB=$PWD/dino_amd/lib/variants/lib_$v.so; fi
  timeout -k 10 120 python tools/bench_mlp3.py $M 20 1 1 2>&1 | grep mlp_fused3 | tail -1
done; done
