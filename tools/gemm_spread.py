import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from dino_amd import capi
from bench_ops import timeit, rand_bf16
lib = capi.lib()
M = 32 * 3601
for name, N, K, epi in (("fc1", 1536, 384, capi.EPI_GELU), ("fc2", 384, 1536, capi.EPI_RESID)):
    A, W = rand_bf16((M, K)), rand_bf16((N, K)); bias = torch.randn(N, device="cuda")
    X = torch.zeros((M, N), device="cuda") if epi == capi.EPI_RESID else None
    O = torch.zeros((M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
    lib.dinoseg_set_option(b"gemm_big", 2 if False else 1)
    def run():
        capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M*K, K, W.data_ptr(), N*K, M, N, K, 1, epi, bias.data_ptr(), capi.ptr(X), capi.ptr(O), M*N, N, capi.stream_ptr()))
    res = {}
    for rnd in range(4):
        for dbg in (0, 4, 1, 5):
            lib.dinoseg_set_option(b"gemm_dbg", dbg)
            res.setdefault(dbg, []).append(timeit(run, iters=8, warm=2))
    print(name, {("burst" if not d & 4 else "spread") + ("/no-epi" if d & 1 else ""): round(sorted(v)[len(v)//2]*1e3, 1) for d, v in res.items()}, flush=True)
lib.dinoseg_set_option(b"gemm_dbg", 0)
