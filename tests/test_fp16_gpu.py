"""fp16 operand format (precision 'fp16', round 4): the single-plane kernels on fp16 operands through the C-ABI (-m gpu).

Same MFMA rate as bf16 (v_mfma_f32_32x32x16_f16), 11 significand bits instead of 8: the linears and Q.K^T run on fp16, the
probabilities and V (the P.V product) stay bf16 -- 2^S against the fixed reference 0 needs bf16's exponent range -- the head
runs split (bf16 hi+lo).  Op tests: against fp64 on the operands the kernel saw, like their bf16 twins in
test_ops_gpu.py (the stand-alone ops take the format from option `op_fmt`).  Model tests: against the goldens captured from the
reference (vision_transformer.py:237-248, pl_torch_modules.py:239-256); the bounds are 1.5x what was measured on MI355X.
"""
import contextlib
import os

import numpy as np
import pytest
import torch

from dino_amd import DINOSeg, ViTConfig, capi, procedural_state_dict
from dino_amd.weights import synthetic_frames, synthetic_labels
from oracle import dinoseg_oracle as O
from tests.gpu_util import pack, pack_slabs, seeded

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634
FP16_MAX = 65504.0


S = capi.stream_ptr


@contextlib.contextmanager
def fp16_ops():
    """Stand-alone ops (dinoseg_op_*) with single-plane fp16 operands."""
    capi.check(capi.lib().dinoseg_set_option(b"op_fmt", 1))
    try:
        yield
    finally:
        capi.check(capi.lib().dinoseg_set_option(b"op_fmt", 0))


def q16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


def qb16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def un16(p: torch.Tensor) -> torch.Tensor:
    return p.view(torch.float16).to(torch.float32).sum(dim=0)


def unb16(p: torch.Tensor) -> torch.Tensor:
    return p.view(torch.bfloat16).to(torch.float32).sum(dim=0)


def _ln_ref(X, gam, bet, eps=1e-6):
    return O.layer_norm(X.cpu(), gam.cpu(), bet.cpu(), eps)


def test_pack_is_round_to_nearest_fp16(cuda):
    x = seeded((37, 50), 1) * 3.0
    x[0, 0], x[0, 1], x[0, 2] = 1e-7, 65000.0, -3.0e-5          # subnormal range, near the top, subnormal
    with fp16_ops():
        p = pack(x, 1, 64, 64)
    assert torch.equal(p.view(torch.float16)[0, :37, :50], x.to(torch.float16))
    assert torch.all(p[0, 37:] == 0) and torch.all(p[0, :, 50:] == 0)
    with fp16_ops():      # hi + lo planes: ~22 bits (the lo plane of the subnormal-range entries underflows: absolute error <= 2^-25)
        p2 = pack(x, 2)
    rec = un16(p2)
    assert float(((rec - x).abs() - 2.0 ** -22 * x.abs()).max()) <= 2.0 ** -25
    assert torch.equal(p2.view(torch.float16)[0], x.to(torch.float16))


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (1, 128, 384), (515, 384, 1536), (3000, 384, 384), (25613, 1152, 384)])
def test_gemm_resid_and_gelu(cuda, M, N, K):
    """C = A W^T on fp16 operands, both GEMM kernels (N % 384 == 0 with >= 128 tiles takes the persistent 256x384 one)."""
    A = seeded((M, K), 10 + M) + torch.arange(K, device="cuda", dtype=torch.float32)[None, :] * 1e-3
    W = seeded((N, K), 20 + N) * 0.1 + torch.arange(N, device="cuda", dtype=torch.float32)[:, None] * 1e-3
    bias = seeded((N,), 3)
    lib = capi.lib()
    X0 = seeded((M, N), 4)
    base = (q16(A).double() @ q16(W).double().t() + bias.double()).float()
    scale = float(base.abs().max())
    for big in (0, 2):
        if big == 2 and (N % 384 != 0 or K % 32 != 0):
            continue
        capi.check(lib.dinoseg_set_option(b"gemm_big", big))
        try:
            with fp16_ops():
                Ap, Wp = pack(A, 1), pack(W, 1)
                X = X0.clone()
                capi.check(lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, 1, capi.EPI_RESID, bias.data_ptr(),
                                               X.data_ptr(), None, 0, 0, S()))
                outp = torch.zeros((1, M, N), dtype=torch.int16, device="cuda")
                capi.check(lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, 1, capi.EPI_GELU, bias.data_ptr(),
                                               None, outp.data_ptr(), M * N, N, S()))
                torch.cuda.synchronize()
        finally:
            capi.check(lib.dinoseg_set_option(b"gemm_big", 1))
        # same quantised operands, fp32 accumulation: only summation-order noise remains
        assert float((X - X0 - base).abs().max()) <= 3e-6 * scale * max(1.0, (K / 64) ** 0.5), big
        want = O.gelu_erf(base.cpu()).cuda()
        got = un16(outp)
        assert torch.isfinite(got).all()
        # fp16 rounding of the stored activation (2^-11) + the fitted GELU of the single-plane modes (2.6e-5)
        assert float((got - want).abs().max()) <= 2.0 ** -11 * float(want.abs().max()) + 1e-4, big
    # the training-only epilogues have no fp16 form: refused, not mis-computed
    with fp16_ops():
        Ap, Wp = pack(A, 1), pack(W, 1)
        out = torch.zeros((M, N), device="cuda")
        rc = lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, 1, capi.EPI_PLAIN, bias.data_ptr(), out.data_ptr(),
                                 None, 0, 0, S())
    assert rc != 0 and "fp16" in capi.last_error()


@pytest.mark.parametrize("route", ["gemm_small", "gemm_big", "ln_gemm"])
@pytest.mark.parametrize("B,ntok", [(2, 197), (3, 130)])
def test_qkv_layout_q_k_fp16_v_bf16(cuda, route, B, ntok):
    """The qkv projection on fp16 operands: Q (pre-scaled) and K leave as fp16, V as bf16; pad rows stay zero."""
    H, D = 6, 384
    npad, M_ = (ntok + 63) // 64 * 64, B * ntok
    lib = capi.lib()
    W, bias = seeded((3 * D, D), 24) * 0.1, seeded((3 * D,), 25)
    plane = B * H * npad * 64
    q = torch.zeros((1, B, H, npad, 64), dtype=torch.int16, device="cuda")
    k, v = torch.zeros_like(q), torch.zeros_like(q)
    qscale = 0.125 * LOG2E
    if route == "ln_gemm":
        X = seeded((M_, D), 21) * 2.0 - 0.3
        gam, bet = 1 + 0.2 * seeded((D,), 22), 0.1 * seeded((D,), 23)
        with fp16_ops():
            Ws = pack_slabs(W, 1)
            capi.check(lib.dinoseg_op_ln_gemm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Ws.data_ptr(), 3 * D * D, bias.data_ptr(),
                                              M_, 3 * D, D, 1, 4, None, 0, q.data_ptr(), k.data_ptr(), v.data_ptr(), plane, ntok, npad, H,
                                              qscale, None, None, S()))
        A = q16(_ln_ref(X, gam, bet).cuda())
        slack = 2.0          # the kernel's own fp32 LayerNorm can move an fp16 rounding point of A
    else:
        A0 = seeded((M_, D), 5)
        capi.check(lib.dinoseg_set_option(b"gemm_big", 2 if route == "gemm_big" else 0))
        try:
            with fp16_ops():
                Ap, Wp = pack(A0, 1), pack(W, 1)
                capi.check(lib.dinoseg_op_qkv_gemm(Ap.data_ptr(), M_ * D, Wp.data_ptr(), 3 * D * D, bias.data_ptr(), B, ntok, npad, H, 1,
                                                   qscale, q.data_ptr(), k.data_ptr(), v.data_ptr(), plane, S()))
        finally:
            capi.check(lib.dinoseg_set_option(b"gemm_big", 1))
        A = q16(A0)
        slack = 1.0
    torch.cuda.synchronize()
    ref = (A.double() @ q16(W).double().t() + bias.double()).float().reshape(B, ntok, 3, H, 64).permute(2, 0, 3, 1, 4)
    gq, gk, gv = un16(q), un16(k), unb16(v)
    top = float(ref.abs().max())
    assert float((gq[:, :, :ntok] - ref[0] * qscale).abs().max()) <= slack * 2.0 ** -11 * top + 1e-4
    assert float((gk[:, :, :ntok] - ref[1]).abs().max()) <= slack * 2.0 ** -11 * top + 1e-4
    assert float((gv[:, :, :ntok] - ref[2]).abs().max()) <= 2.0 ** -8 * top + 1e-4          # bf16
    # ... and V really is bf16 (its low mantissa bits are zero as fp32 -> bf16 -> fp32 round trips exactly)
    assert torch.equal(qb16(gv), gv)
    assert torch.all(q[:, :, :, ntok:] == 0) and torch.all(k[:, :, :, ntok:] == 0) and torch.all(v[:, :, :, ntok:] == 0)


@pytest.mark.parametrize("M_,N", [(394, 768), (128 * 300 + 77, 1536)])
def test_ln_gemm_gelu(cuda, M_, N):
    """LayerNorm fused into the fc1 GEMM (gemm_ln12.hip) on fp16 operands: out = fp16(gelu(LN(X) W^T + b))."""
    K = 384
    X = seeded((M_, K), 11) * 1.7 + 0.4
    gam, bet = 1 + 0.2 * seeded((K,), 12), 0.1 * seeded((K,), 13)
    W, bias = seeded((N, K), 14) * 0.1, seeded((N,), 15)
    out = torch.zeros((1, M_, N), dtype=torch.int16, device="cuda")
    lib = capi.lib()
    with fp16_ops():
        Ws = pack_slabs(W, 1)
        capi.check(lib.dinoseg_op_ln_gemm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Ws.data_ptr(), N * K, bias.data_ptr(), M_, N, K,
                                          1, capi.EPI_GELU, out.data_ptr(), M_ * N, None, None, None, 0, 0, 0, 6, 0.0, None, None, S()))
        # the by-products of the training forward have no fp16 form
        aout = torch.zeros((1, M_, K), dtype=torch.int16, device="cuda")
        rc = lib.dinoseg_op_ln_gemm(X.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Ws.data_ptr(), N * K, bias.data_ptr(), M_, N, K, 1,
                                    capi.EPI_GELU, out.data_ptr(), M_ * N, None, None, None, 0, 0, 0, 6, 0.0, aout.data_ptr(), None, S())
    assert rc != 0 and "inference-only" in capi.last_error()
    torch.cuda.synchronize()
    A = q16(_ln_ref(X, gam, bet))
    z = (A.double() @ q16(W).cpu().double().t() + bias.cpu().double()).float()
    want = O.gelu_erf(z)
    got = un16(out).cpu()
    assert torch.isfinite(got).all()
    # one fp16 rounding point of A moved by the kernel's fp32 LayerNorm changes one of 384 products by 2^-11 of itself
    assert float((got - want).abs().max()) <= 2.0 ** -10 * float(want.abs().max()) + 3e-4


@pytest.mark.parametrize("B,H,ntok", [(1, 1, 64), (2, 2, 197), (1, 3, 65), (1, 2, 3601)])
def test_attention_fp16_qk(cuda, B, H, ntok):
    """attention_z.hip with fp16 Q / K (the QK^T product on v_mfma_f32_32x32x16_f16), bf16 V and probabilities, fp16 ctx."""
    npad = (ntok + 63) // 64 * 64
    g = np.random.default_rng(ntok)
    Q = torch.from_numpy(g.standard_normal((B, H, ntok, 64)).astype(np.float32)) * 1.5 * (0.125 * LOG2E)
    K = torch.from_numpy(g.standard_normal((B, H, ntok, 64)).astype(np.float32)) * 1.5
    V = torch.from_numpy(g.standard_normal((B, H, ntok, 64)).astype(np.float32))

    def padded(x):
        full = torch.zeros((B, H, npad, 64), dtype=torch.float32)
        full[:, :, :ntok] = x
        return full.reshape(-1, 64).cuda()

    with fp16_ops():
        qp, kp = pack(padded(Q), 1), pack(padded(K), 1)
    vp = pack(padded(V), 1)                                  # bf16
    ctx = torch.zeros((1, B * ntok, H * 64), dtype=torch.int16, device="cuda")
    with fp16_ops():
        capi.check(capi.lib().dinoseg_op_attention(qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), B * H * npad * 64, ctx.data_ptr(),
                                                   B * ntok * H * 64, None, B, H, ntok, npad, 1, S()))
        lse = torch.zeros((B, H, ntok), dtype=torch.float32, device="cuda")
        rc = capi.lib().dinoseg_op_attention(qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), B * H * npad * 64, ctx.data_ptr(),
                                             B * ntok * H * 64, lse.data_ptr(), B, H, ntok, npad, 1, S())
    assert rc != 0 and "inference-only" in capi.last_error()       # no log-sum-exp: there is no fp16 backward
    torch.cuda.synchronize()
    qq = un16(qp).reshape(B, H, npad, 64)[:, :, :ntok].double().cpu() / LOG2E
    kk = un16(kp).reshape(B, H, npad, 64)[:, :, :ntok].double().cpu()
    vv = unb16(vp).reshape(B, H, npad, 64)[:, :, :ntok].double().cpu()
    ref = (torch.softmax(qq @ kk.transpose(-1, -2), dim=-1) @ vv).transpose(1, 2).reshape(B * ntok, H * 64).float()
    got = un16(ctx).cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 1.2e-2          # bf16 probabilities, as in the bf16 mode


@pytest.mark.parametrize("M_", [77, 128 * 5 + 33, 128 * 300 + 19])
def test_proj_mlp_fused(cuda, M_):
    """mlp_fused2.hip (projection + MLP in one launch) on fp16 operands: against fp64 on the operands the kernel sees (fp16 ctx /
    weights, fp16 LayerNorm and GELU outputs), and the MLP-only launch against the same."""
    D_, F_ = 384, 1536
    X = seeded((M_, D_), 41) * 1.7 + 0.4 + torch.arange(D_, device="cuda", dtype=torch.float32)[None, :] * 1e-3
    ctx = q16(seeded((M_, D_), 48) * 0.8)
    Wpr = seeded((D_, D_), 49) * 0.07 + torch.arange(D_, device="cuda", dtype=torch.float32)[:, None] * 1e-5
    bpr = seeded((D_,), 50) * 0.3
    gam, bet = 1 + 0.2 * seeded((D_,), 42), 0.1 * seeded((D_,), 43)
    W1 = seeded((F_, D_), 44) * 0.06 + torch.arange(F_, device="cuda", dtype=torch.float32)[:, None] * 1e-5
    b1 = seeded((F_,), 45) * 0.5
    W2 = seeded((D_, F_), 46) * 0.04 + torch.arange(D_, device="cuda", dtype=torch.float32)[:, None] * 1e-5
    b2 = seeded((D_,), 47)
    lib = capi.lib()
    got, got_mlp = X.clone(), X.clone()
    with fp16_ops():
        Wp = torch.empty((2 * D_ * F_,), dtype=torch.int16, device="cuda")
        capi.check(lib.dinoseg_op_pack_mlp(W1.data_ptr(), W2.data_ptr(), D_, F_, Wp.data_ptr(), S()))
        Wprp = torch.empty((D_ * D_,), dtype=torch.int16, device="cuda")
        capi.check(lib.dinoseg_op_pack_proj(Wpr.data_ptr(), D_, Wprp.data_ptr(), S()))
        ctx_h = pack(ctx, 1)
        capi.check(lib.dinoseg_op_proj_mlp_fused(got.data_ptr(), ctx_h.data_ptr(), Wprp.data_ptr(), bpr.data_ptr(), gam.data_ptr(),
                                                 bet.data_ptr(), 1e-6, Wp.data_ptr(), b1.data_ptr(), b2.data_ptr(), M_, D_, F_, S()))
        capi.check(lib.dinoseg_op_mlp_fused(got_mlp.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-6, Wp.data_ptr(), b1.data_ptr(),
                                            b2.data_ptr(), M_, D_, F_, S()))
    torch.cuda.synchronize()

    def mlp64(x64):
        A = q16(_ln_ref(x64.float(), gam, bet).cuda()).double()
        z = A @ q16(W1).double().t() + b1.double()
        Hq = q16(O.gelu_erf(z.float().cpu()).cuda()).double()
        return Hq @ q16(W2).double().t() + b2.double()

    xmid = X.double() + ctx.double() @ q16(Wpr).double().t() + bpr.double()
    delta = mlp64(xmid)
    scale = float(delta.abs().max())
    assert torch.isfinite(got).all()
    # fp32 summation order, the fitted GELU (2.6e-5), fp16 rounding points moved by the kernel's own fp32 arithmetic
    assert float((got - (xmid + delta).float()).abs().max()) <= 2.0 ** -11 * scale + 1e-3
    assert float((got - X).abs().max()) > 0.5 * scale
    d2 = mlp64(X.double())
    assert float((got_mlp - (X.double() + d2).float()).abs().max()) <= 2.0 ** -11 * float(d2.abs().max()) + 1e-3


# ------------------------------------------------------------------------------------------------ whole model
def build(cfg, precision):
    if isinstance(cfg, int):
        cfg = ViTConfig(n_blocks=cfg)
    sd = procedural_state_dict(cfg)
    m = DINOSeg(head=cfg.head, n_blocks=cfg.n_blocks, n_classes=cfg.n_classes, precision=precision, arch=cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to("cuda:0"), sd, cfg


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


# measured on MI355X (round 4) over the five routes: L=1 2.50e-2 / 5 flips, L=12 2.15e-2 .. 2.43e-2 / 4-8 flips of 3600 (with the patch
# embedding on bf16 hi+lo planes, option fp16_patch_planes = 2: 2.24e-2 .. 2.64e-2 / 5-8); the CPU emulation of the mix,
# oracle/precision_ablation.py: 2.2e-2 / 6; the bf16 mode: 0.12-0.14 / 15-24.  Bounds = 1.5x
FP16_BOUND = {1: (3.8e-2, 9), 12: (4.0e-2, 12)}


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("L", [1, 12])
def test_g3_vits8_480_one_wave_fused_mlp_is_bounded(cuda, golden_dir, L, precision):
    """Library option mlp_fused4: the projection + MLP half of every block through mlp_fused4.hip (one wave per SIMD) instead of
    mlp_fused2.hip -- against the reference's log-probabilities (G3) within the mode's bounds, and close to the default route's output."""
    g = load(golden_dir, f"g3_vits8_L{L}_r480")
    m, _, _ = build(L, precision)
    frames = torch.from_numpy(synthetic_frames(1, 480, seed=int(g["frame_seed"]))).cuda()
    lib = capi.lib()
    out = {}
    try:
        for f4 in (1, 0):      # (the option is read when the weights are packed -- at the first forward of this model -- and at every forward)
            for k, v in (("mlp_fused", 2), ("proj_fused", 1), ("qkv_fused", 0), ("mlp_fused4", f4)):
                capi.check(lib.dinoseg_set_option(k.encode(), v))
            lp, am = m.forward_frames(frames)
            torch.cuda.synchronize()
            out[f4] = (lp.cpu(), am.cpu().long())
    finally:
        for k, v in (("mlp_fused", 1), ("proj_fused", 1), ("qkv_fused", 0), ("mlp_fused4", 0)):
            capi.check(lib.dinoseg_set_option(k.encode(), v))
    lp, am = out[1]
    assert torch.isfinite(lp).all()
    err = float((lp - torch.from_numpy(g["logp"])).abs().max())
    flips = int((am != torch.from_numpy(g["argmax"].astype(np.int64))).sum())
    tol, max_flips = FP16_BOUND[L] if precision == "fp16" else (0.2, 36)
    print(f"{precision} L={L} mlp_fused4: max|dlogp| {err:.3e}, {flips} flips; against the two-wave kernel {float((lp - out[0][0]).abs().max()):.3e}")
    assert err <= tol and flips <= max_flips, (err, flips)
    assert not torch.equal(lp, out[0][0])          # (the option took effect: another summation order)
    assert float((lp - out[0][0]).abs().max()) <= tol


@pytest.mark.parametrize("mlp_fused,proj_fused,qkv_fused,gemm_ln", [(0, 0, 0, 0), (0, 0, 0, 2), (2, 0, 0, 2), (2, 1, 0, 2), (2, 1, 1, 2)])
@pytest.mark.parametrize("L", [1, 12])
def test_g3_vits8_480_fp16_mode_is_bounded(cuda, golden_dir, L, mlp_fused, proj_fused, qkv_fused, gemm_ln):
    """The fp16 mode against the reference's log-probabilities (G3), on every dispatch route of the linears (separate LayerNorm +
    GEMMs, LayerNorm-fused GEMMs, fused MLP, + projection, + qkv tail)."""
    g = load(golden_dir, f"g3_vits8_L{L}_r480")
    m, _, _ = build(L, "fp16")
    frames = torch.from_numpy(synthetic_frames(1, 480, seed=int(g["frame_seed"]))).cuda()
    lib = capi.lib()
    for k, v in (("mlp_fused", mlp_fused), ("proj_fused", proj_fused), ("qkv_fused", qkv_fused), ("gemm_ln", gemm_ln)):
        capi.check(lib.dinoseg_set_option(k.encode(), v))
    try:
        lp, am = m.forward_frames(frames)
        torch.cuda.synchronize()
    finally:
        for k, v in (("mlp_fused", 1), ("proj_fused", 1), ("qkv_fused", 0), ("gemm_ln", 1)):
            capi.check(lib.dinoseg_set_option(k.encode(), v))
    assert torch.isfinite(lp).all()
    err = float((lp.cpu() - torch.from_numpy(g["logp"])).abs().max())
    flips = int((am.cpu().long() != torch.from_numpy(g["argmax"].astype(np.int64))).sum())
    tol, max_flips = FP16_BOUND[L]
    print(f"fp16 L={L} routes={mlp_fused}{proj_fused}{qkv_fused}{gemm_ln}: max|dlogp| {err:.3e}, {flips} flips")
    assert err <= tol and flips <= max_flips, (err, flips)
    # every flipped patch sits inside the error band of its top-2 margin
    bad = am.cpu().long() != torch.from_numpy(g["argmax"].astype(np.int64))
    assert float(torch.from_numpy(g["margin"])[bad].max() if bad.any() else 0.0) <= 2 * tol


def test_fp16_is_closer_to_the_reference_than_bf16_at_batch_32(cuda, golden_dir):
    """The point of the mode: a full batch through the large-batch dispatch (fused kernels, two streams), fp16 against bf16."""
    g = load(golden_dir, "g3_vits8_L12_r480")
    one = synthetic_frames(1, 480, seed=int(g["frame_seed"]))
    frames = torch.from_numpy(np.repeat(one, 12, axis=0)).cuda()
    ref = torch.from_numpy(g["logp"])
    errs = {}
    for prec in ("fp16", "bf16"):
        m, _, _ = build(12, prec)
        lp, _ = m.forward_frames(frames)
        lp = lp.cpu().reshape(12, 3600, -1)
        assert torch.isfinite(lp).all()
        assert float((lp - lp[0:1]).abs().max()) == 0.0          # the copies of the frame agree bit for bit (both halves, every item)
        errs[prec] = float((lp[0] - ref).abs().max())
    print("batch 12 max|dlogp|:", errs)
    assert errs["fp16"] <= FP16_BOUND[12][0] and errs["fp16"] < 0.4 * errs["bf16"]


def test_g7_vitb8_fp16(cuda, golden_dir):
    """ViT-B/8 (embed_dim 768: separate LayerNorm + the persistent GEMMs on fp16 operands) against the reference rows of G7."""
    g = load(golden_dir, "g7_vitb8_L12_r480")
    from dino_amd.weights import VIT_B8
    cfg = ViTConfig(embed_dim=VIT_B8.embed_dim, num_heads=VIT_B8.num_heads, n_blocks=12)
    m, _, _ = build(cfg, "fp16")
    frames = torch.from_numpy(synthetic_frames(1, 480, seed=int(g["frame_seed"]))).cuda()
    lp, am = m.forward_frames(frames)
    rows = torch.from_numpy(g["rows"])
    err = float((lp.cpu()[rows] - torch.from_numpy(g["logp_rows"])).abs().max())
    print(f"fp16 ViT-B/8: max|dlogp| {err:.3e}")
    assert torch.isfinite(lp).all() and err <= 1.2e-2          # measured 7.7e-3 (bf16: 4.5e-2)


@pytest.mark.parametrize("chan,head", [(40.0, 5.0), (100.0, 8.0)])
def test_outlier_channels_do_not_overflow_fp16(cuda, chan, head):
    """The weights of test_model_gpu.py::test_outlier_channels_and_sharp_heads (three residual channels x chan = "massive
    activations", one head's q / k rows x head: its scores leave the zero-reference kernel's fast range) in the fp16 mode: nothing
    on the path leaves fp16's range (no inf / NaN in the output or in the residual stream after any block), and the result stays
    closer to the CPU oracle than the bf16 mode's."""
    from tests.test_model_gpu import _outlier_state
    cfg = ViTConfig(n_blocks=3)
    sd = _outlier_state(procedural_state_dict(cfg), cfg, chan, head)
    frames_np = synthetic_frames(2, 112, seed=33)
    with torch.no_grad():
        ref = O.dinoseg_forward(O.preprocess(frames_np), O.to_torch(sd), cfg.num_heads)
    errs = {}
    for prec in ("fp16", "bf16"):
        m = DINOSeg(head=cfg.head, n_blocks=3, n_classes=cfg.n_classes, precision=prec, arch=cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        m = m.to("cuda:0")
        lp, _ = m.forward_frames(torch.from_numpy(frames_np).cuda())
        assert torch.isfinite(lp).all(), prec
        errs[prec] = float((lp.cpu() - ref).abs().max())
        if prec == "fp16":
            x = O.preprocess(frames_np).cuda()
            for blk in range(4):
                assert torch.isfinite(m.debug_tokens(x, blk)).all(), blk
    print(f"outliers x{chan:g} / x{head:g}: max|dlogp| {errs}")
    assert errs["fp16"] < errs["bf16"]


def test_fp16_small_paths_and_refusals(cuda, golden_dir):
    """The side paths in fp16 -- single frame (small-batch dispatch), get_last_selfattention, forward_mask, features -- against the
    parity mode; the fine-tune entry points refuse the precision."""
    cfg = ViTConfig(n_blocks=3)
    m16, _, _ = build(cfg, "fp16")
    m3, _, _ = build(cfg, "bf16x3")
    x = O.preprocess(synthetic_frames(1, 96, seed=102)).cuda()
    a16, a3 = m16.get_last_selfattention(x), m3.get_last_selfattention(x)
    assert float((a16 - a3).abs().max()) <= 5e-3 and float((a16.sum(-1) - 1).abs().max()) <= 1e-5
    t16, t3 = m16.dino(x), m3.dino(x)
    assert float((t16 - t3).abs().max()) <= 3e-2
    masks = (torch.rand(3, 12, 12, generator=torch.Generator().manual_seed(1)) > 0.5).float()
    e16, e3 = m16.forward_mask(x, masks), m3.forward_mask(x, masks)
    assert torch.isfinite(e16).all() and float((e16 - e3).abs().max()) <= 3e-2
    frames = torch.from_numpy(synthetic_frames(2, 96, seed=5)).cuda()
    labels = torch.from_numpy(synthetic_labels(2, 144, cfg.n_classes, seed=6)).cuda()
    m16.unfreeze_bb()
    with pytest.raises(capi.DinosegError, match="inference-only"):
        m16.fused_training_step((frames, labels), 0)
    with pytest.raises(capi.DinosegError, match="inference-only"):
        torch.nn.functional.nll_loss(m16(frames), labels.reshape(-1)).backward()


# ================================================================================================ fp16 hi + lo planes ('fp16x3')
# Three MFMAs per product like bf16x3, on fp16 planes: ~22 significand bits instead of ~16.  Everything is fp16 here (V, the
# probabilities, the patch embedding and the head included); GEMM outputs saturate at +-65504.
def test_gemm_hi_lo_planes_fp16(cuda):
    """C = A W^T on fp16 hi + lo planes, both GEMM kernels: against the fp32 operands' exact product (22 bits: 2^-5 of bf16x3's
    error), RESID / RELU / PATCH-free epilogues; saturation of an out-of-range output."""
    M, N, K = 515, 384, 1536
    A = seeded((M, K), 10) + torch.arange(K, device="cuda", dtype=torch.float32)[None, :] * 1e-3
    W = seeded((N, K), 20) * 0.1 + torch.arange(N, device="cuda", dtype=torch.float32)[:, None] * 1e-3
    bias = seeded((N,), 3)
    lib = capi.lib()
    ref = (A.double() @ W.double().t() + bias.double()).float()
    scale = float(ref.abs().max())
    X0 = seeded((M, N), 4)
    for big in (0, 2):
        capi.check(lib.dinoseg_set_option(b"gemm_big", big))
        try:
            with fp16_ops():
                Ap, Wp = pack(A, 2), pack(W, 2)
                X = X0.clone()
                capi.check(lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, 2, capi.EPI_RESID, bias.data_ptr(),
                                               X.data_ptr(), None, 0, 0, S()))
                outp = torch.zeros((2, M, N), dtype=torch.int16, device="cuda")
                capi.check(lib.dinoseg_op_gemm(Ap.data_ptr(), M * K, K, Wp.data_ptr(), N * K, M, N, K, 2, capi.EPI_RELU, bias.data_ptr(),
                                               None, outp.data_ptr(), M * N, N, S()))
                torch.cuda.synchronize()
        finally:
            capi.check(lib.dinoseg_set_option(b"gemm_big", 1))
        err = float((X - X0 - ref).abs().max())
        assert err <= 3e-6 * scale, (big, err / scale)          # (bf16 hi + lo: 4e-5; fp32 summation noise at K = 1536 is ~1e-6)
        got = un16(outp)
        assert float((got - torch.relu(ref)).abs().max()) <= 3e-6 * scale + 2.0 ** -24
    # an output beyond the fp16 range saturates (both planes finite) instead of becoming inf - inf
    with fp16_ops():
        Ah, Wh = pack(A * 300.0, 2), pack(W * 300.0, 2)
        outp = torch.zeros((2, M, N), dtype=torch.int16, device="cuda")
        capi.check(lib.dinoseg_op_gemm(Ah.data_ptr(), M * K, K, Wh.data_ptr(), N * K, M, N, K, 2, capi.EPI_RELU, bias.data_ptr(), None,
                                       outp.data_ptr(), M * N, N, S()))
    got = un16(outp)
    assert torch.isfinite(got).all() and float(got.max()) == FP16_MAX and float((ref * 9e4).max()) > FP16_MAX


@pytest.mark.parametrize("B,H,ntok", [(2, 2, 197), (1, 2, 3601)])
def test_attention_hi_lo_planes_fp16(cuda, B, H, ntok):
    """attention.hip on fp16 hi + lo planes (Q, K, V, the probabilities and ctx): against fp64 on the operands."""
    npad = (ntok + 63) // 64 * 64
    g = np.random.default_rng(ntok + 7)
    Q = torch.from_numpy(g.standard_normal((B, H, ntok, 64)).astype(np.float32)) * 1.5 * (0.125 * LOG2E)
    K = torch.from_numpy(g.standard_normal((B, H, ntok, 64)).astype(np.float32)) * 1.5
    V = torch.from_numpy(g.standard_normal((B, H, ntok, 64)).astype(np.float32))
    K[:, :, ntok - 3] = Q[:, :, 5] / (0.125 * LOG2E) * 4.0          # one dominant key late in the sweep: the deferred rescale runs

    def padded(x):
        full = torch.zeros((B, H, npad, 64), dtype=torch.float32)
        full[:, :, :ntok] = x
        return full.reshape(-1, 64).cuda()

    ctx = torch.zeros((2, B * ntok, H * 64), dtype=torch.int16, device="cuda")
    with fp16_ops():
        qp, kp, vp = pack(padded(Q), 2), pack(padded(K), 2), pack(padded(V), 2)
        capi.check(capi.lib().dinoseg_op_attention(qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), B * H * npad * 64, ctx.data_ptr(),
                                                   B * ntok * H * 64, None, B, H, ntok, npad, 2, S()))
    torch.cuda.synchronize()
    qq = un16(qp).reshape(B, H, npad, 64)[:, :, :ntok].double().cpu() / LOG2E
    kk = un16(kp).reshape(B, H, npad, 64)[:, :, :ntok].double().cpu()
    vv = un16(vp).reshape(B, H, npad, 64)[:, :, :ntok].double().cpu()
    ref = (torch.softmax(qq @ kk.transpose(-1, -2), dim=-1) @ vv).transpose(1, 2).reshape(B * ntok, H * 64).float()
    got = un16(ctx).cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 1e-5          # (bf16 hi + lo: 1e-4)


@pytest.mark.parametrize("L", [1, 3, 12])
def test_g3_vits8_480_fp16x3_parity(cuda, golden_dir, L):
    """The north-star bar (|dlogp| <= 1e-3, argmax identical) in the fp16 hi + lo mode, with margin: measured .. (bf16x3: 2.2e-4 .. 3.8e-4)."""
    g = load(golden_dir, f"g3_vits8_L{L}_r480")
    m, _, _ = build(L, "fp16x3")
    frames = torch.from_numpy(synthetic_frames(1, 480, seed=int(g["frame_seed"]))).cuda()
    lp, am = m.forward_frames(frames)
    err = float((lp.cpu() - torch.from_numpy(g["logp"])).abs().max())
    print(f"fp16x3 L={L}: max|dlogp| {err:.3e}")
    assert err <= 3e-4
    assert torch.equal(am.cpu().long(), torch.from_numpy(g["argmax"].astype(np.int64)))


def test_fp16x3_other_goldens_and_batch(cuda, golden_dir):
    """@960 (G4 L=12, 14 401 tokens), ViT-B/8 (G7), and a batch of 9 frames @480 through the large-batch routes (LayerNorm + the hi+lo
    configuration of the persistent GEMM, two streams): bar 1e-3 / identical argmax everywhere, every copy of the frame bit-identical."""
    g = load(golden_dir, "g4_vits8_L12_r960")
    m, _, _ = build(12, "fp16x3")
    lp, am = m.forward_frames(torch.from_numpy(synthetic_frames(1, 960, seed=int(g["frame_seed"]))).cuda())
    rows = torch.from_numpy(g["rows"])
    e960 = float((lp.cpu()[rows] - torch.from_numpy(g["logp_rows"])).abs().max())
    differ = am.cpu().numpy() != g["argmax"].astype(np.int32)
    lpc = lp.cpu()
    top2 = lpc.topk(2, dim=1).values
    print(f"fp16x3 @960: max|dlogp| {e960:.3e}, flips {int(differ.sum())} / 14400, reference margins of the flipped patches "
          f"{g['margin'][differ]}, own margins {(top2[:, 0] - top2[:, 1])[torch.from_numpy(differ)].numpy()}")
    # argmax identical except across a TIE of the reference itself: a patch whose two best classes differ by less than the reference's
    # own fp32 rounding noise at this depth (fp32 against fp64 evaluation of the same graph: ~3e-5) has no defined winner
    assert e960 <= 3e-4 and np.all(g["margin"][differ] <= 6e-5) and int(differ.sum()) <= 2
    g3 = load(golden_dir, "g3_vits8_L12_r480")
    one = synthetic_frames(1, 480, seed=int(g3["frame_seed"]))
    lp, am = m.forward_frames(torch.from_numpy(np.repeat(one, 9, axis=0)).cuda())
    lp = lp.cpu().reshape(9, 3600, -1)
    assert float((lp - lp[0:1]).abs().max()) == 0.0
    e9 = float((lp[0] - torch.from_numpy(g3["logp"])).abs().max())
    assert e9 <= 3e-4 and torch.equal(am.cpu().long().reshape(9, 3600)[8], torch.from_numpy(g3["argmax"].astype(np.int64)))
    g7 = load(golden_dir, "g7_vitb8_L12_r480")
    from dino_amd.weights import VIT_B8
    mb, _, _ = build(ViTConfig(embed_dim=VIT_B8.embed_dim, num_heads=VIT_B8.num_heads, n_blocks=12), "fp16x3")
    lp, am = mb.forward_frames(torch.from_numpy(synthetic_frames(1, 480, seed=int(g7["frame_seed"]))).cuda())
    eb = float((lp.cpu()[torch.from_numpy(g7["rows"])] - torch.from_numpy(g7["logp_rows"])).abs().max())
    print(f"fp16x3: @960 {e960:.3e}, batch 9 @480 {e9:.3e}, ViT-B/8 {eb:.3e}")
    assert eb <= 2e-4


@pytest.mark.parametrize("chan,head", [(40.0, 5.0), (100.0, 8.0)])
def test_fp16x3_holds_the_flat_bar_under_outliers(cuda, chan, head):
    """the weights of test_outlier_channels_and_sharp_heads (x40 / x5 and x100 / x8) against the CPU oracle -- where
    bf16 hi + lo planes sit AT the 1e-3 bar (9.9e-4 .. 1.36e-3, route-dependent), fp16 hi + lo planes hold it flat with margin
    (CPU emulation: 1.4e-4 / 1.6e-4), on both dispatch routes of the small batch."""
    from tests.test_model_gpu import _outlier_state
    cfg = ViTConfig(n_blocks=3)
    sd = _outlier_state(procedural_state_dict(cfg), cfg, chan, head)
    frames_np = synthetic_frames(2, 112, seed=33)
    with torch.no_grad():
        ref = O.dinoseg_forward(O.preprocess(frames_np), O.to_torch(sd), cfg.num_heads)
    m = DINOSeg(head=cfg.head, n_blocks=3, n_classes=cfg.n_classes, precision="fp16x3", arch=cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0")
    lib = capi.lib()
    for big in (1, 0):
        capi.check(lib.dinoseg_set_option(b"gemm_big", big))
        try:
            lp, am = m.forward_frames(torch.from_numpy(frames_np).cuda())
            torch.cuda.synchronize()
        finally:
            capi.check(lib.dinoseg_set_option(b"gemm_big", 1))
        assert torch.isfinite(lp).all()
        err = float((lp.cpu() - ref).abs().max())
        print(f"outliers x{chan:g} / x{head:g} fp16x3 (gemm_big={big}): max|dlogp| {err:.3e}")
        assert err <= 1e-3 / 2          # half the north-star bar
        top2 = ref.topk(2, dim=1).values
        flips = am.cpu().long() != ref.argmax(1)
        assert not bool((flips & ((top2[:, 0] - top2[:, 1]) > 1e-3)).any())


def test_fp16x3_side_paths_and_refusal(cuda):
    cfg = ViTConfig(n_blocks=3)
    m16, _, _ = build(cfg, "fp16x3")
    m3, _, _ = build(cfg, "bf16x3")
    x = O.preprocess(synthetic_frames(1, 96, seed=102)).cuda()
    assert float((m16.get_last_selfattention(x) - m3.get_last_selfattention(x)).abs().max()) <= 2e-4
    assert float((m16.dino(x) - m3.dino(x)).abs().max()) <= 2e-3
    masks = (torch.rand(3, 12, 12, generator=torch.Generator().manual_seed(1)) > 0.5).float()
    assert float((m16.forward_mask(x, masks) - m3.forward_mask(x, masks)).abs().max()) <= 2e-3
    frames = torch.from_numpy(synthetic_frames(2, 96, seed=5)).cuda()
    labels = torch.from_numpy(synthetic_labels(2, 144, cfg.n_classes, seed=6)).cuda()
    m16.unfreeze_bb()
    with pytest.raises(capi.DinosegError, match="inference-only"):
        m16.fused_training_step((frames, labels), 0)
