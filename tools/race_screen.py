"""Race screen for the hand-synchronised kernels (counted vmcnt, LDS-DMA rings, asm loads): every configuration is run
many times on identical inputs, with other kernels in flight on a second stream to perturb timing, and every output must be
bit-identical to the first run and (for the GEMMs) agree with the other GEMM kernel.  python tools/race_screen.py [repeats]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from dino_amd import capi
from gpu_util import pack, pack_slabs, quant_like, seeded, unpack
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
    # (planes 2: the 128 x 384 hi + lo configuration -- its 16-bit epilogue pairs the third column blocks of neighbouring waves through their
    #  LDS patches behind raw workgroup barriers, like the single-plane one)
    for planes, epi, ename in ((1, capi.EPI_PLAIN, "plain"), (1, capi.EPI_RESID, "resid"), (1, capi.EPI_GELU, "gelu"), (2, capi.EPI_GELU, "gelu x3")):
        if planes == 2 and M * N > 60_000_000:
            continue
        Ap, Wp = pack(A, planes), pack(W, planes)
        capi.check(lib.dinoseg_set_option(b"gemm_big", 2))
        X0 = seeded((M, N), 4)
        out = torch.zeros((M, N), device="cuda")
        g = torch.zeros((planes, M, N), dtype=torch.int16, device="cuda")

        def run():
            if epi == capi.EPI_RESID:
                out.copy_(X0)
            capi.check(lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, planes, epi, bias.data_ptr(),
                                           out.data_ptr() if epi != capi.EPI_GELU else None, g.data_ptr() if epi == capi.EPI_GELU else None,
                                           M * N, N, S()))
        screen(f"gemm_big {ename:7s} M={M} N={N} K={K}", run, lambda: [g] if epi == capi.EPI_GELU else [out])
        # agreement with the 128x128 kernel
        big = (unpack(g) if epi == capi.EPI_GELU else out).clone()
        capi.check(lib.dinoseg_set_option(b"gemm_big", 0))
        run(); torch.cuda.synchronize()
        small = unpack(g) if epi == capi.EPI_GELU else out
        err = float((big - small).abs().max()) / max(float(small.abs().max()), 1e-6)
        tol = (2.0 ** -7 if planes == 1 else 2.0 ** -14) if epi == capi.EPI_GELU else 3e-5
        print(f"    vs 128x128 kernel: rel max diff {err:.2e} {'OK' if err <= tol else 'MISMATCH'}", flush=True)
        bad += err > tol
    del A, W, Ap, Wp
capi.check(lib.dinoseg_set_option(b"gemm_big", 1))

# LayerNorm-fused A-stationary GEMM (gemm_ln.hip): private LDS-DMA rings with counted vmcnt, inline-asm LDS reads with counted
# lgkmcnt, two barriers per panel; also against LayerNorm + the 128x128 kernel on the same operands
LOG2E = 1.4426950408889634
for (Bq, ntok, planes) in [(32, 3601, 1), (8, 3601, 2), (3, 197, 1), (1, 65, 2), (5, 1031, 1)]:
    Hh, D = 6, 384
    M, npad = Bq * ntok, (ntok + 63) // 64 * 64
    X = seeded((M, D), 31) * 1.3 + 0.2
    gam, bet = 1 + 0.2 * seeded((D,), 32), 0.1 * seeded((D,), 33)
    for name, N, epi in (("qkv", 3 * D, 4), ("fc1", 4 * D, capi.EPI_GELU)):
        W, bias = seeded((N, D), 34) * 0.1, seeded((N,), 35)
        Ws = pack_slabs(W, planes)
        out = torch.zeros((planes, M, N), dtype=torch.int16, device="cuda")
        q = torch.zeros((planes, Bq, Hh, npad, 64), dtype=torch.int16, device="cuda")
        k, vt = torch.zeros_like(q), torch.zeros_like(q)

        def run():
            capi.check(lib.dinoseg_op_ln_gemm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Ws.data_ptr(), 0, bias.data_ptr(), M, N, D,
                                              planes, epi, out.data_ptr(), M * N, q.data_ptr(), k.data_ptr(), vt.data_ptr(),
                                              Bq * Hh * npad * 64, ntok, npad, Hh, 0.125 * LOG2E, None, None, S()))
        screen(f"gemm_ln {name} B={Bq} N={ntok} planes={planes}", run, (lambda: [q, k, vt]) if epi == 4 else (lambda: [out]))
        if epi != 4:
            A = torch.zeros((planes, M, D), dtype=torch.int16, device="cuda")
            capi.check(lib.dinoseg_op_layernorm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, M, D, A.data_ptr(), M * D, planes, None, 0,
                                                ntok, S()))
            Wp = pack(W, planes)
            ref = torch.zeros_like(out)
            capi.check(lib.dinoseg_set_option(b"gemm_big", 0))
            capi.check(lib.dinoseg_op_gemm(A.data_ptr(), M * D, D, Wp.data_ptr(), N * D, M, N, D, planes, epi, bias.data_ptr(), None,
                                           ref.data_ptr(), M * N, N, S()))
            capi.check(lib.dinoseg_set_option(b"gemm_big", 1))
            err = float((unpack(out) - unpack(ref)).abs().max()) / max(float(unpack(ref).abs().max()), 1e-6)
            tol = 2.0 ** -6 if planes == 1 else 2.0 ** -13
            print(f"    vs LayerNorm + 128x128 kernel: rel max diff {err:.2e} {'OK' if err <= tol else 'MISMATCH'}", flush=True)
            bad += err > tol

import test_ops_gpu as T
# (round 5: 11 | 1024 [| 65536] = the assembly tile loop of attention_za.hip, 32 / 64 queries per wave; | 2048 = at every grid size;
#  planes = 2 with bits 4 / 10 = its hi + lo body)
for variant in (3, 11, 11 | 1024 | 2048, 11 | 1024 | 2048 | 65536, 11 | 16 | 1024 | 2048):
  capi.check(lib.dinoseg_set_option(b"attn_variant", variant))
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
      screen(f"attention variant {variant} B={B} H={H} N={ntok} planes={planes}", run, lambda: [ctx, lse])
capi.check(lib.dinoseg_set_option(b"attn_variant", 11 | 1024 | 65536))

# round 6: the one-wave-per-SIMD fused launches (three-slot LDS-DMA ring shared by four waves, counted vmcnt over pieces, row loads and stores in ONE
# in-order queue, counted lgkmcnt over the fragment reads, M0 set once per four pieces) and the row-stationary GEMMs (same ring)
D_, F_, H_ = 384, 1536, 6
for (Bq, ntok, fp16) in [(32, 3601, True), (11, 3601, False), (3, 130, True), (1, 65, False)]:
    M_ = Bq * ntok
    npad = (ntok + 63) // 64 * 64
    c = T._mlp3_case(M_, fp16, 300, tail=True)
    ctx_pl, _ = T._split_planes(c["ctx"], fp16)
    x3 = torch.zeros_like(c["X"])
    q3 = torch.zeros((2, Bq, H_, npad, 64), dtype=torch.int16, device="cuda")
    k3, v3 = torch.zeros_like(q3), torch.zeros_like(q3)

    def run3():
        x3.copy_(c["X"])
        capi.check(lib.dinoseg_op_proj_mlp_fused3(x3.data_ptr(), ctx_pl.data_ptr(), M_ * D_, c["bpr"].data_ptr(), 1e-6, c["Wp"].data_ptr(),
                                                  c["b2"].data_ptr(), M_, D_, F_, int(fp16), S()))
    screen(f"mlp_fused3 B={Bq} N={ntok} fp16={fp16}", run3, lambda: [x3])

    def run3t():
        x3.copy_(c["X"])
        capi.check(lib.dinoseg_op_block_tail_fused3(x3.data_ptr(), ctx_pl.data_ptr(), M_ * D_, c["bpr"].data_ptr(), 1e-6, c["Wp"].data_ptr(),
                                                    c["b2"].data_ptr(), q3.data_ptr(), k3.data_ptr(), v3.data_ptr(), Bq * H_ * npad * 64, Bq, ntok,
                                                    npad, H_, 0.125 * LOG2E, int(fp16), D_, F_, int(fp16), S()))
    screen(f"mlp_fused3 + qkv tail B={Bq} N={ntok} fp16={fp16}", run3t, lambda: [x3, q3, k3, v3])
    ctx1, _ = T._one_plane(c["ctx"], fp16)
    Wp4 = T._pack_mlp4(c["Wpr"], c["W1"], c["b1"], c["W2"], c["gam"], c["bet"], fp16)
    x4 = torch.zeros_like(c["X"])

    def run4():
        x4.copy_(c["X"])
        capi.check(lib.dinoseg_op_proj_mlp_fused4(x4.data_ptr(), ctx1.data_ptr(), c["bpr"].data_ptr(), 1e-6, Wp4.data_ptr(), c["b2"].data_ptr(), M_, D_,
                                                  F_, int(fp16), S()))
    screen(f"mlp_fused4 B={Bq} N={ntok} fp16={fp16}", run4, lambda: [x4])
    Wp4t = T._pack_mlp4(c["Wpr"], c["W1"], c["b1"], c["W2"], c["gam"], c["bet"], fp16, tail=(c["Wqkv"], c["bq"], c["gam1"], c["bet1"]))
    q4 = torch.zeros((Bq, H_, npad, 64), dtype=torch.int16, device="cuda")
    k4, v4 = torch.zeros_like(q4), torch.zeros_like(q4)

    def run4t():
        x4.copy_(c["X"])
        capi.check(lib.dinoseg_op_block_tail_fused4(x4.data_ptr(), ctx1.data_ptr(), c["bpr"].data_ptr(), 1e-6, Wp4t.data_ptr(), c["b2"].data_ptr(),
                                                    q4.data_ptr(), k4.data_ptr(), v4.data_ptr(), Bq, ntok, npad, H_, 0.125 * LOG2E, D_, F_, int(fp16), S()))
    screen(f"mlp_fused4 + qkv tail B={Bq} N={ntok} fp16={fp16}", run4t, lambda: [x4, q4, k4, v4])
    del c, ctx_pl, x3, q3, k3, v3, x4, Wp4, Wp4t, q4, k4, v4

for (M_, fp16) in [(57616, True), (24001, False), (300, True)]:
    capi.check(lib.dinoseg_set_option(b"op_fmt", int(fp16)))
    K = 768
    Ap, _ = T._one_plane(seeded((M_, K), 61) * 0.7, fp16)
    for name, N, kind, epi in (("fc1 gelu", 3072, 0, capi.EPI_GELU), ("proj resid", 768, 1, capi.EPI_RESID)):
        W, bias = seeded((N, K), 62) * 0.05, seeded((N,), 63) * 0.3
        Wp = T._pack_rs(W, kind)
        X0 = seeded((M_, N), 64) if epi == capi.EPI_RESID else None
        xo = torch.zeros((M_, N), device="cuda") if epi == capi.EPI_RESID else None
        o16 = torch.zeros((M_, N), dtype=torch.int16, device="cuda") if epi == capi.EPI_GELU else None

        def runrs():
            if xo is not None:
                xo.copy_(X0)
            capi.check(lib.dinoseg_op_gemm_rs(Ap.data_ptr(), K, Wp.data_ptr(), bias.data_ptr(), M_, N, K, epi, None if xo is None else xo.data_ptr(),
                                              None if o16 is None else o16.data_ptr(), N if o16 is not None else 0, None, None, None, 0, 0, 0, 0.0, S()))
        screen(f"gemm_rs {name} M={M_} fp16={fp16}", runrs, lambda: [xo if xo is not None else o16])
capi.check(lib.dinoseg_set_option(b"op_fmt", 0))
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
