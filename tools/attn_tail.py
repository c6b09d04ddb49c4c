"""How much of the attention launch is tail?  Same (B, H) at N = 3584 (28 full q-tiles: 5376 workgroups = 7.0 rounds of
768 slots) and N = 3601 (29 q-tiles, the last with 17 valid rows: 5568 workgroups = 7.25 rounds)."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from dino_amd import capi
from bench_ops import rand_bf16, timeit
lib = capi.lib()
B, H = 32, 6
res = {}
for rnd in range(3):
    for ntok in (3584, 3601, 3648):
        npad = (ntok + 63) // 64 * 64
        q, k, v = (rand_bf16((1, B, H, npad, 64)) for _ in range(3))
        ctx = torch.zeros((1, B * ntok, H * 64), dtype=torch.int16, device="cuda")
        def run():
            capi.check(lib.dinoseg_op_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), B * H * npad * 64, ctx.data_ptr(),
                                                B * ntok * H * 64, None, B, H, ntok, npad, 1, capi.stream_ptr()))
        res.setdefault(ntok, []).append(timeit(run, iters=10, warm=3))
for ntok, ts in res.items():
    t = sorted(ts)[len(ts) // 2]
    print(f"N={ntok}: {t * 1e3:7.1f} us   {4.0 * B * H * ntok * ntok * 64 / (t * 1e-3) / 1e12:6.1f} TFLOP/s   per N^2: {t * 1e3 / ntok / ntok * 1e6:.3f}")
