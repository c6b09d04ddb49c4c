import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from dino_amd import capi
from bench_ops import timeit, rand_bf16
lib = capi.lib()
B, ntok, D = 32, 3601, 384
M = B * ntok
shapes = [("qkv", 3 * D, D, "qkv"), ("fc1", 4 * D, D, capi.EPI_GELU), ("fc2", D, 4 * D, capi.EPI_RESID)]
npad = 3648
for name, N, K, epi in shapes:
    A, W = rand_bf16((M, K)), rand_bf16((N, K)); bias = torch.randn(N, device="cuda")
    X = torch.zeros((M, N), device="cuda") if epi == capi.EPI_RESID else None
    O = torch.zeros((M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
    if epi == "qkv":
        q = torch.zeros((B, 6, npad, 64), dtype=torch.int16, device="cuda"); k = torch.zeros_like(q); v = torch.zeros_like(q)
    def run():
        if epi == "qkv":
            capi.check(lib.dinoseg_op_qkv_gemm(A.data_ptr(), M*K, W.data_ptr(), N*K, bias.data_ptr(), B, ntok, npad, 6, 1, 0.18, q.data_ptr(), k.data_ptr(), v.data_ptr(), B*6*npad*64, capi.stream_ptr()))
        else:
            capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M*K, K, W.data_ptr(), N*K, M, N, K, 1, epi, bias.data_ptr(), capi.ptr(X), capi.ptr(O), M*N, N, capi.stream_ptr()))
    res = {}
    for rnd in range(4):
        for st in (0, 1, 2, 3, 4, 6):
            lib.dinoseg_set_option(b"gemm_stagger", st)
            res.setdefault(st, []).append(timeit(run, iters=8, warm=2))
    print(name, {st: round(sorted(v)[len(v)//2]*1e3, 1) for st, v in res.items()}, flush=True)
lib.dinoseg_set_option(b"gemm_stagger", 0)
