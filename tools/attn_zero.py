"""Clock / power check: the same attention launch on random and on all-zero operands (zeros draw less power,
so the chip holds a higher clock for an identical instruction stream: MI355X_MICROARCH.md 'DVFS give-back')."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from dino_amd import capi
from bench_ops import timeit, rand_bf16
lib = capi.lib()
B, H, ntok, npad = 32, 6, 3601, 3648
ctx = torch.zeros((1, B * ntok, H * 64), dtype=torch.int16, device="cuda")
data = {"random": [rand_bf16((1, B, H, npad, 64)) for _ in range(3)], "zeros": [torch.zeros((1, B, H, npad, 64), dtype=torch.int16, device="cuda") for _ in range(3)]}
res = {}
for rnd in range(4):
    for name, (q, k, v) in data.items():
        def run():
            capi.check(lib.dinoseg_op_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), B*H*npad*64, ctx.data_ptr(), B*ntok*H*64, None, B, H, ntok, npad, 1, capi.stream_ptr()))
        res.setdefault(name, []).append(timeit(run, iters=8, warm=2))
for k, v in sorted(res.items()):
    print(k, "median %.1f us" % (sorted(v)[len(v)//2]*1e3))
