"""Race screen for the hand-synchronised kernels (counted vmcnt, LDS-DMA rings, asm loads): every configuration is run
many times on identical inputs, with other kernels in flight on a second stream to perturb timing, and every output must be
bit-identical to the first run and (for the GEMMs) agree with the other GEMM kernel.  python tools/race_screen.py [repeats]"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from dino_amd import capi
from gpu_util import pack, seeded, unpack
lib = capi.lib()
S = capi.stream_ptr
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device="cuda")
bad = 0


def perturb():
    with torch.cuda.stream(side):
        for _ in range(2):
            noise_a.mul_(1.0001)


def screen(name, run, outs):
    global bad
    run(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs()]
    nd = 0
    for i in range(REP):
        if i % 3 == 0:
            perturb()
        run()
        torch.cuda.synchronize()
        if any(not torch.equal(a, b) for a, b in zip(ref, outs())):
            nd += 1
    print(f"{name:58s} {'OK' if nd == 0 else 'DIFFERS in %d runs' % nd}", flush=True)
    bad += nd


for (M, N, K) in [(115232, 384, 384), (115232, 1536, 384), (28808, 384, 1536), (3000, 1152, 384), (257, 384, 64), (70001, 384, 1536)]:
    A, W, bias = seeded((M, K), 1), seeded((N, K), 2) * 0.1, seeded((N,), 3)
    Ap, Wp = pack(A, 1), pack(W, 1)
    for epi, ename in ((capi.EPI_PLAIN, "plain"), (capi.EPI_RESID, "resid"), (capi.EPI_GELU, "gelu")):
        capi.check(lib.dinoseg_set_option(b"gemm_big", 2))
        X0 = seeded((M, N), 4)
        out = torch.zeros((M, N), device="cuda")
        g = torch.zeros((1, M, N), dtype=torch.int16, device="cuda")

        def run():
            if epi == capi.EPI_RESID:
                out.copy_(X0)
            capi.check(lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, 1, epi, bias.data_ptr(),
                                           out.data_ptr() if epi != capi.EPI_GELU else None, g.data_ptr() if epi == capi.EPI_GELU else None,
                                           M * N, N, S()))
        screen(f"gemm_big {ename:5s} M={M} N={N} K={K}", run, lambda: [g] if epi == capi.EPI_GELU else [out])
        # agreement with the 128x128 kernel
        big = (unpack(g) if epi == capi.EPI_GELU else out).clone()
        capi.check(lib.dinoseg_set_option(b"gemm_big", 0))
        run(); torch.cuda.synchronize()
        small = unpack(g) if epi == capi.EPI_GELU else out
        err = float((big - small).abs().max()) / max(float(small.abs().max()), 1e-6)
        tol = 2.0 ** -7 if epi == capi.EPI_GELU else 3e-5
        print(f"    vs 128x128 kernel: rel max diff {err:.2e} {'OK' if err <= tol else 'MISMATCH'}", flush=True)
        bad += err > tol
    del A, W, Ap, Wp
capi.check(lib.dinoseg_set_option(b"gemm_big", 1))

import test_ops_gpu as T
for (B, H, ntok, planes) in [(32, 6, 3601, 1), (2, 6, 3601, 2), (4, 2, 197, 1), (3, 3, 64, 1), (1, 1, 129, 2), (8, 6, 14401, 1)]:
    npad = (ntok + 63) // 64 * 64
    q = (torch.randn((planes, B, H, npad, 64), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    k = (torch.randn((planes, B, H, npad, 64), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    v = (torch.randn((planes, B, H, npad, 64), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    k[..., ntok:, :] = 0; v[..., ntok:, :] = 0
    ctx = torch.zeros((planes, B * ntok, H * 64), dtype=torch.int16, device="cuda")
    lse = torch.zeros((B, H, ntok), device="cuda")

    def run():
        capi.check(lib.dinoseg_op_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), B * H * npad * 64, ctx.data_ptr(),
                                            B * ntok * H * 64, lse.data_ptr(), B, H, ntok, npad, planes, S()))
    screen(f"attention B={B} H={H} N={ntok} planes={planes}", run, lambda: [ctx, lse])
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
