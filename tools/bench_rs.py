"""GPU box: the row-stationary GEMMs (gemm_rs.hip) against the persistent 256 x 384 kernel (gemm_big.hip) on ViT-B/8's four linears,
16 frames @480 (57 616 rows), fp16 operands:  python tools/bench_rs.py [rows] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_amd import capi  # noqa: E402
from tests.gpu_util import seeded  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 3601
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = capi.lib()
S = capi.stream_ptr
capi.check(lib.dinoseg_set_option(b"op_fmt", 1))
D, H, ntok = 768, 12, 3601
B = M // ntok
npad = (ntok + 63) // 64 * 64


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, N, K, epi in (("qkv", 3 * D, D, 4), ("proj", D, D, capi.EPI_RESID), ("fc1", 4 * D, D, capi.EPI_GELU), ("fc2", D, 4 * D, capi.EPI_RESID)):
    A = (seeded((M, K), 1) * 0.5).to(torch.float16).view(torch.int16)
    W = seeded((N, K), 2) * 0.03
    Wq = W.to(torch.float16).view(torch.int16).contiguous()
    bias = seeded((N,), 3) * 0.1
    Wp = torch.empty((N * K,), dtype=torch.int16, device="cuda")
    capi.check(lib.dinoseg_op_pack_rs(W.data_ptr(), N, K, 1 if epi == capi.EPI_RESID else 0, Wp.data_ptr(), S()))
    X = torch.zeros((M, N), device="cuda") if epi == capi.EPI_RESID else None
    O = torch.zeros((M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
    q = torch.zeros((B, H, npad, 64), dtype=torch.int16, device="cuda")
    k, v = torch.zeros_like(q), torch.zeros_like(q)

    def run_rs():
        capi.check(lib.dinoseg_op_gemm_rs(A.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), M, N, K, epi, capi.ptr(X), capi.ptr(O), N,
                                          q.data_ptr(), k.data_ptr(), v.data_ptr(), ntok, npad, H, 0.18, S()))

    def run_big():
        if epi == 4:
            capi.check(lib.dinoseg_op_qkv_gemm(A.data_ptr(), M * K, Wq.data_ptr(), N * K, bias.data_ptr(), B, ntok, npad, H, 1, 0.18,
                                               q.data_ptr(), k.data_ptr(), v.data_ptr(), B * H * npad * 64, S()))
        else:
            capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M * K, K, Wq.data_ptr(), N * K, M, N, K, 1, epi, bias.data_ptr(), capi.ptr(X), capi.ptr(O),
                                           M * N, N, S()))
    for rep in range(2):
        t_rs, t_big = timeit(run_rs), timeit(run_big)
        fl = 2.0 * M * N * K
        print(f"{name:5s} N={N:5d} K={K:5d}: gemm_rs {t_rs:7.1f} us {fl / t_rs / 1e6:6.0f} TFLOP/s | gemm_big {t_big:7.1f} us {fl / t_big / 1e6:6.0f} TFLOP/s", flush=True)
capi.check(lib.dinoseg_set_option(b"op_fmt", 0))
