"""GPU box: time the single-plane fused projection + MLP launches alone, one wave per SIMD (mlp_fused4.hip) against two (mlp_fused2.hip):
    python tools/bench_mlp4.py [rows] [iters] [fp16: 1|0] [proj: 1|0] [which: 4|2|42]
DINOSEG_LIB selects the build for A/B runs (ablation builds: make EXTRA=-DMF4_ABL=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_amd import capi  # noqa: E402
from tests.gpu_util import seeded  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 3601
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fp16 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
proj = int(sys.argv[4]) if len(sys.argv) > 4 else 1
which = sys.argv[5] if len(sys.argv) > 5 else "42"
D, F = 384, 1536
dt = torch.float16 if fp16 else torch.bfloat16
X = seeded((M, D), 1) * 1.5
gam, bet = 1 + 0.2 * seeded((D,), 2), 0.1 * seeded((D,), 3)
W1, b1 = seeded((F, D), 4) * 0.06, seeded((F,), 5) * 0.5
W2, b2 = seeded((D, F), 6) * 0.002, seeded((D,), 7) * 0.01
Wpr, bpr = seeded((D, D), 9) * 0.01, seeded((D,), 10) * 0.01
ctx = (seeded((M, D), 8) * 0.5).to(dt).contiguous().view(torch.int16)
lib = capi.lib()
S = capi.stream_ptr
capi.check(lib.dinoseg_set_option(b"op_fmt", fp16))
Wp4 = torch.zeros((lib.dinoseg_op_mlp4_pack_elems(D, F),), dtype=torch.int16, device="cuda")
Wq, bq = seeded((3 * D, D), 11) * 0.05, seeded((3 * D,), 12) * 0.1
capi.check(lib.dinoseg_op_pack_mlp4(Wpr.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), gam.data_ptr(), bet.data_ptr(), Wq.data_ptr(), bq.data_ptr(),
                                    gam.data_ptr(), bet.data_ptr(), D, F, fp16, Wp4.data_ptr(), S()))
Wp2 = torch.zeros((lib.dinoseg_op_mlp_fused_pack_elems(D, F),), dtype=torch.int16, device="cuda")
capi.check(lib.dinoseg_op_pack_mlp(W1.data_ptr(), W2.data_ptr(), D, F, Wp2.data_ptr(), S()))
Wpr2 = torch.zeros((lib.dinoseg_op_proj_pack_elems(D),), dtype=torch.int16, device="cuda")
capi.check(lib.dinoseg_op_pack_proj(Wpr.data_ptr(), D, Wpr2.data_ptr(), S()))


def run4():
    capi.check(lib.dinoseg_op_proj_mlp_fused4(X.data_ptr(), ctx.data_ptr() if proj else None, bpr.data_ptr(), 1e-6, Wp4.data_ptr(), b2.data_ptr(), M, D, F,
                                              fp16, S()))


def run2():
    if proj:
        capi.check(lib.dinoseg_op_proj_mlp_fused(X.data_ptr(), ctx.data_ptr(), Wpr2.data_ptr(), bpr.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                                 1e-6, Wp2.data_ptr(), b1.data_ptr(), b2.data_ptr(), M, D, F, S()))
    else:
        capi.check(lib.dinoseg_op_mlp_fused(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Wp2.data_ptr(), b1.data_ptr(), b2.data_ptr(),
                                            M, D, F, S()))


for rep in range(3):
    for name, run in (("mlp_fused4", run4), ("mlp_fused2", run2)):
        if name[-1] not in which:
            continue
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        fl = 2.0 * M * D * (2 * F + (D if proj else 0))
        print(f"{name} fp16={fp16} proj={proj} M={M}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s"
              f"  lib={os.path.basename(os.environ.get('DINOSEG_LIB', 'in-tree'))}", flush=True)
assert torch.isfinite(X).all()
