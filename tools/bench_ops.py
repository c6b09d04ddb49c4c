#!/usr/bin/env python3
"""Operator micro-benchmarks on the GPU box (called through the C-ABI, timed with events on the launch stream).

    python tools/bench_ops.py gemm      # the four DINOSeg GEMM shapes, both kernels, with timing ablations
    python tools/bench_ops.py attn      # fused attention at the benchmark shape
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dino_amd import capi  # noqa: E402

LOG2E = 1.4426950408889634


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def rand_bf16(shape):
    return (torch.randn(shape, device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)


def bench_gemm(quick=True):
    lib = capi.lib()
    B, ntok, D = 32, 3601, 384
    M = B * ntok
    shapes = [("qkv", 3 * D, D, "qkv"), ("proj", D, D, capi.EPI_RESID), ("fc1", 4 * D, D, capi.EPI_GELU),
              ("fc2", D, 4 * D, capi.EPI_RESID)]
    npad = (ntok + 63) // 64 * 64
    for name, N, K, epi in shapes:
        A, W = rand_bf16((M, K)), rand_bf16((N, K))
        bias = torch.randn(N, device="cuda")
        X = torch.zeros((M, N), device="cuda") if epi == capi.EPI_RESID else None
        O = torch.zeros((M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
        if epi == "qkv":
            q = torch.zeros((B, 6, npad, 64), dtype=torch.int16, device="cuda")
            k, vt = torch.zeros_like(q), torch.zeros_like(q)
        for big in (0, 1, 2):
            for dbg in ((0, 1) if quick else (0, 1, 2, 3)):
                capi.check(lib.dinoseg_set_option(b"gemm_big", big))
                capi.check(lib.dinoseg_set_option(b"gemm_dbg", dbg))

                def run():
                    if epi == "qkv":
                        capi.check(lib.dinoseg_op_qkv_gemm(A.data_ptr(), M * K, W.data_ptr(), N * K, bias.data_ptr(), B, ntok,
                                                           npad, 6, 1, 0.125 * LOG2E, q.data_ptr(), k.data_ptr(), vt.data_ptr(),
                                                           B * 6 * npad * 64, capi.stream_ptr()))
                    else:
                        capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M * K, K, W.data_ptr(), N * K, M, N, K, 1, epi,
                                                       bias.data_ptr(), capi.ptr(X), capi.ptr(O), M * N, N, capi.stream_ptr()))
                ms = timeit(run)
                tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                print(f"{name:5s} N={N:5d} K={K:5d} kernel={('small', 'auto', 'big')[big]:8s} dbg={dbg} "
                      f"(skip epilogue={dbg & 1}, skip loads={(dbg >> 1) & 1}): {ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
    lib.dinoseg_set_option(b"gemm_big", 1)
    lib.dinoseg_set_option(b"gemm_dbg", 0)


def bench_gemm_vitb():
    """the persistent kernel on ViT-B/8's shapes (16 frames @480: 57 616 rows), bf16, with the timing ablations"""
    lib = capi.lib()
    M, D = 16 * 3601, 768
    for name, N, K, epi in [("proj", D, D, capi.EPI_RESID), ("fc2", D, 4 * D, capi.EPI_RESID), ("fc1", 4 * D, D, capi.EPI_GELU)]:
        A, W = rand_bf16((M, K)), rand_bf16((N, K))
        bias = torch.randn(N, device="cuda")
        X = torch.zeros((M, N), device="cuda") if epi == capi.EPI_RESID else None
        O = torch.zeros((M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
        for dbg in (0, 1, 2, 3):
            capi.check(lib.dinoseg_set_option(b"gemm_big", 2))
            capi.check(lib.dinoseg_set_option(b"gemm_dbg", dbg))

            def run():
                capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M * K, K, W.data_ptr(), N * K, M, N, K, 1, epi,
                                               bias.data_ptr(), capi.ptr(X), capi.ptr(O), M * N, N, capi.stream_ptr()))
            ms = timeit(run)
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"vitb {name:5s} N={N:5d} K={K:5d} skip_epilogue={dbg & 1} skip_loads={dbg >> 1}: {ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
        del A, W, X, O
    lib.dinoseg_set_option(b"gemm_big", 1)
    lib.dinoseg_set_option(b"gemm_dbg", 0)


def bench_gemm_planes2():
    """bf16x3 (hi + lo planes): the 128x128 kernel against the 128x384 configuration of the persistent kernel"""
    lib = capi.lib()
    B, ntok, D = 32, 3601, 384
    M = B * ntok
    for name, N, K, epi in [("proj", D, D, capi.EPI_RESID), ("fc2", D, 4 * D, capi.EPI_RESID), ("fc1", 4 * D, D, capi.EPI_GELU)]:
        A, W = rand_bf16((2, M, K)), rand_bf16((2, N, K))
        bias = torch.randn(N, device="cuda")
        X = torch.zeros((M, N), device="cuda") if epi == capi.EPI_RESID else None
        O = torch.zeros((2, M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
        for big in (0, 2):
            for dbg in ((0, 1) if big == 0 else (0, 1, 2, 3)):
                capi.check(lib.dinoseg_set_option(b"gemm_big", big))
                capi.check(lib.dinoseg_set_option(b"gemm_dbg", dbg))

                def run():
                    capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M * K, K, W.data_ptr(), N * K, M, N, K, 2, epi,
                                                   bias.data_ptr(), capi.ptr(X), capi.ptr(O), M * N, N, capi.stream_ptr()))
                ms = timeit(run)
                tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                print(f"{name:5s} planes=2 N={N:5d} K={K:5d} kernel={('small', 'auto', 'big')[big]:6s} skip_epilogue={dbg & 1} skip_loads={dbg >> 1}: "
                      f"{ms * 1e3:8.1f} us  {tf:7.1f} TFLOP/s ({3 * tf:7.1f} MFMA-equivalent)", flush=True)
        del A, W, X, O
    lib.dinoseg_set_option(b"gemm_big", 1)
    lib.dinoseg_set_option(b"gemm_dbg", 0)


def bench_lngemm():
    """LN + GEMM as two launches (layernorm_kernel + gemm_big) against the fused A-stationary kernel (gemm_ln.hip), interleaved."""
    lib = capi.lib()
    B, ntok, D, H = int(os.environ.get("LNGEMM_B", "32")), 3601, 384, 6
    M, npad = B * ntok, (ntok + 63) // 64 * 64
    X = torch.randn((M, D), device="cuda") * 1.5 + 0.2
    gam, bet = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda") * 0.1
    for planes in (1, 2):
        A = torch.zeros((planes, M, D), dtype=torch.int16, device="cuda")
        for name, N, epi in (("qkv", 3 * D, 4), ("fc1", 4 * D, capi.EPI_GELU)):
            W = torch.stack([rand_bf16((N, D)) for _ in range(planes)])
            Wf = torch.randn((N, D), device="cuda") * 0.5
            nslab = lib.dinoseg_op_ln_gemm_slab_elems(N, D, planes)
            Ws = torch.empty((nslab,), dtype=torch.int16, device="cuda")
            capi.check(lib.dinoseg_op_pack_slabs(Wf.data_ptr(), N, D, planes, Ws.data_ptr(), capi.stream_ptr()))
            bias = torch.randn(N, device="cuda")
            out = torch.zeros((planes, M, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None
            q = torch.zeros((planes, B, H, npad, 64), dtype=torch.int16, device="cuda")
            k, vt = torch.zeros_like(q), torch.zeros_like(q)

            def ln():
                capi.check(lib.dinoseg_op_layernorm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, M, D, A.data_ptr(), M * D, planes,
                                                    None, 0, ntok, capi.stream_ptr()))

            def gemm():
                if epi == 4:
                    capi.check(lib.dinoseg_op_qkv_gemm(A.data_ptr(), M * D, W.data_ptr(), N * D, bias.data_ptr(), B, ntok, npad, H, planes,
                                                       0.125 * LOG2E, q.data_ptr(), k.data_ptr(), vt.data_ptr(), B * H * npad * 64,
                                                       capi.stream_ptr()))
                else:
                    capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M * D, D, W.data_ptr(), N * D, M, N, D, planes, epi, bias.data_ptr(), None,
                                                   out.data_ptr(), M * N, N, capi.stream_ptr()))

            def two():
                ln()
                gemm()

            def fused():
                capi.check(lib.dinoseg_op_ln_gemm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Ws.data_ptr(), N * D, bias.data_ptr(), M, N,
                                                  D, planes, epi, capi.ptr(out), M * N, q.data_ptr(), k.data_ptr(), vt.data_ptr(),
                                                  B * H * npad * 64, ntok, npad, H, 0.125 * LOG2E, None, None, capi.stream_ptr()))
            def ablate(bits):
                def f():
                    lib.dinoseg_set_option(b"gemm_dbg", bits)
                    fused()
                    lib.dinoseg_set_option(b"gemm_dbg", 0)
                return f
            cases = [("LN", ln), ("GEMM", gemm), ("LN+GEMM", two), ("fused", fused)]
            if os.environ.get("LNGEMM_ONLY_FUSED"):
                cases = [("fused", fused)]
            else:
                cases += [("no-epi", ablate(1)), ("no-stores", ablate(64)), ("no-dma", ablate(2)), ("no-ln", ablate(4)), ("no-epi-ln", ablate(5)),
                      ("mfma-only", ablate(7)), ("no-mfma", ablate(8)), ("reads+loop only", ablate(15)), ("mfma+loop only", ablate(39)),
                          ("loop only", ablate(47))]
            res = {nm: [] for nm, _ in cases}
            for rnd in range(5):
                for nm, fn in cases:
                    res[nm].append(timeit(fn, iters=8, warm=2))
            fl = 2.0 * M * N * D
            for nm, ts in res.items():
                t = sorted(ts)[len(ts) // 2]
                print(f"{name} planes={planes} {nm:8s}: {t * 1e3:7.1f} us  {fl / (t * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)


ATTN_VARIANTS = os.environ.get("ATTN_VARIANTS", "11").split(",")        # "<attn_variant>" or "<attn_variant>:<attn_dbg>"


def bench_attn():
    """Interleaved A/B rounds in one process (clocks drift by +-10 % between back-to-back measurements)."""
    lib = capi.lib()
    B, H, ntok = int(os.environ.get("ATTN_B", "32")), 6, int(os.environ.get("ATTN_NTOK", "3601"))
    npad = (ntok + 63) // 64 * 64
    base = dict(attn_variant=3)
    variants = [(f"variant {v}", dict(base, attn_variant=int(v.split(":")[0]), attn_dbg=int((v + ":0").split(":")[1]))) for v in ATTN_VARIANTS]
    op_fmt = int(os.environ.get("OP_FMT", "0"))       # 1: fp16 Q / K (single plane), as precision 'fp16' runs the kernel
    capi.check(lib.dinoseg_set_option(b"op_fmt", op_fmt))
    for planes in [int(v) for v in os.environ.get("ATTN_PLANES", "1,2").split(",")]:
        q = rand_bf16((planes, B, H, npad, 64))
        k = rand_bf16((planes, B, H, npad, 64))
        if op_fmt == 1:
            q = (torch.randn((planes, B, H, npad, 64), device="cuda") * 0.5).to(torch.float16).view(torch.int16)
            k = (torch.randn((planes, B, H, npad, 64), device="cuda") * 0.5).to(torch.float16).view(torch.int16)
        vt = rand_bf16((planes, B, H, npad, 64))
        vt[..., ntok:, :] = 0
        ctx = torch.zeros((planes, B * ntok, H * 64), dtype=torch.int16, device="cuda")

        def run():
            capi.check(lib.dinoseg_op_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), B * H * npad * 64, ctx.data_ptr(),
                                                B * ntok * H * 64, None, B, H, ntok, npad, planes, capi.stream_ptr()))
        times = {name: [] for name, _ in variants}
        for rnd in range(6):
            for name, opts in variants:
                for kk, vv in opts.items():
                    capi.check(lib.dinoseg_set_option(kk.encode(), vv))
                times[name].append(timeit(run, iters=8, warm=2))
        fl = 4.0 * B * H * ntok * ntok * 64
        for name, _ in variants:
            t = sorted(times[name][1:])
            print(f"attention B={B} ntok={ntok} planes={planes} {name:18s}: min {t[0] * 1e3:7.1f} us  median {t[len(t) // 2] * 1e3:7.1f} us  "
                  f"{fl / (t[len(t) // 2] * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)
    for kk, vv in base.items():
        lib.dinoseg_set_option(kk.encode(), vv)
    lib.dinoseg_set_option(b"attn_dbg", 0)
    lib.dinoseg_set_option(b"op_fmt", 0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    {"gemm": bench_gemm, "gemm2": bench_gemm_planes2, "gemmb": bench_gemm_vitb, "attn": bench_attn, "lngemm": bench_lngemm}[what]()
