import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from dino_amd import capi
from bench_ops import timeit, rand_bf16
lib = capi.lib()
M, N, K = 32*3601, 1536, 384
A, W = rand_bf16((M, K)), rand_bf16((N, K)); bias = torch.randn(N, device="cuda")
O = torch.zeros((M, N), dtype=torch.int16, device="cuda")
res = {}
for rnd in range(4):
    for big in (0, 1):
        for epi, name in ((capi.EPI_GELU, "gelu"), (capi.EPI_RELU, "relu")):
            lib.dinoseg_set_option(b"gemm_big", big)
            def run():
                capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M*K, K, W.data_ptr(), N*K, M, N, K, 1, epi, bias.data_ptr(), None, O.data_ptr(), M*N, N, capi.stream_ptr()))
            res.setdefault((big, name), []).append(timeit(run, iters=8, warm=2))
for k, v in sorted(res.items()):
    print(k, "median %.1f us" % (sorted(v)[len(v)//2]*1e3))
