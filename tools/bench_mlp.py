"""GPU box: time the fused MLP operator alone: python tools/bench_mlp.py [rows] [iters] [variant]
(variant 2 = mlp_fused2.hip (1 is accepted for old command lines: the same kernel), 3 = mlp_fused2.hip with the attention output projection in the same launch, 4 = ... and LayerNorm1 + qkv of the next block, 0 = the
separate projection GEMM (dinoseg_op_gemm EPI_RESID); DINOSEG_LIB selects the build for A/B runs of ablation variants)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_amd import capi  # noqa: E402
from tests.gpu_util import seeded  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 3601
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3]
D, F = 384, 1536
X = seeded((M, D), 1) * 1.5
gam, bet = 1 + 0.2 * seeded((D,), 2), 0.1 * seeded((D,), 3)
W1, b1 = seeded((F, D), 4) * 0.06, seeded((F,), 5) * 0.5
W2, b2 = seeded((D, F), 6) * 0.002, seeded((D,), 7) * 0.01
n = capi.lib().dinoseg_op_mlp_fused_pack_elems(D, F)
Wp = torch.empty((n,), dtype=torch.int16, device="cuda")
S = capi.stream_ptr
capi.check(capi.lib().dinoseg_op_pack_mlp(W1.data_ptr(), W2.data_ptr(), D, F, Wp.data_ptr(), S()))


from tests.gpu_util import pack  # noqa: E402
ctx = pack(seeded((M, D), 8) * 0.5, 1)
Wpr, bpr = seeded((D, D), 9) * 0.01, seeded((D,), 10) * 0.01
Wprp = torch.empty((capi.lib().dinoseg_op_proj_pack_elems(D),), dtype=torch.int16, device="cuda")
capi.check(capi.lib().dinoseg_op_pack_proj(Wpr.data_ptr(), D, Wprp.data_ptr(), S()))
Wq = pack(Wpr, 1)
NB = M // 3601      # mode 4 (projection + MLP + LayerNorm1 + qkv of the next block): whole frames of 3601 tokens
Wqkv, bq = seeded((3 * D, D), 11) * 0.05, seeded((3 * D,), 12) * 0.1
Wqp = torch.empty((capi.lib().dinoseg_op_qkv_pack_elems(D),), dtype=torch.int16, device="cuda")
capi.check(capi.lib().dinoseg_op_pack_qkv(Wqkv.data_ptr(), D, Wqp.data_ptr(), S()))
qb = torch.zeros((max(NB, 1), 6, 3648, 64), dtype=torch.int16, device="cuda")
kb, vb = torch.zeros_like(qb), torch.zeros_like(qb)
mode = 2


def run():
    lib = capi.lib()
    if mode == 3:
        capi.check(lib.dinoseg_op_proj_mlp_fused(X.data_ptr(), ctx.data_ptr(), Wprp.data_ptr(), bpr.data_ptr(), gam.data_ptr(),
                                                 bet.data_ptr(), 1e-6, Wp.data_ptr(), b1.data_ptr(), b2.data_ptr(), M, D, F, S()))
    elif mode == 4:
        capi.check(lib.dinoseg_op_block_tail_fused(X.data_ptr(), ctx.data_ptr(), Wprp.data_ptr(), bpr.data_ptr(), gam.data_ptr(),
                                                   bet.data_ptr(), 1e-6, Wp.data_ptr(), b1.data_ptr(), b2.data_ptr(), Wqp.data_ptr(),
                                                   bq.data_ptr(), gam.data_ptr(), bet.data_ptr(), qb.data_ptr(), kb.data_ptr(), vb.data_ptr(),
                                                   NB, 3601, 3648, 6, 0.18, D, F, S()))
    elif mode == 0:
        capi.check(lib.dinoseg_op_gemm(ctx.data_ptr(), M * D, D, Wq.data_ptr(), D * D, M, D, D, 1, capi.EPI_RESID, bpr.data_ptr(),
                                       X.data_ptr(), None, 0, D, S()))
    else:
        capi.check(lib.dinoseg_op_mlp_fused(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Wp.data_ptr(), b1.data_ptr(),
                                            b2.data_ptr(), M, D, F, S()))


for rep in range(2):
    for v in variants:
        mode = v
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            run()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        fl = 2 * 2 * M * D * F
        print(f"mlp_fused variant {v} M={M}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s  lib={os.path.basename(os.environ.get('DINOSEG_LIB', 'in-tree'))}")
