"""GPU box: random (resolution, batch) shapes through every dispatch route of the linears, all against one reference route of the same
model (bounded: the routes differ only in rounding points / summation order), two streams bit-identical to one.
    python tools/fuzz_routes.py [cases] [seed] [precision: bf16 (default) | fp16 | bf16x3 | fp16x3]
single-plane precisions: separate kernels, fused MLP, + projection, + qkv tail, the one-wave fused launch, one / two streams; hi+lo precisions:
128x128 kernel only, persistent GEMM wherever allowed, LayerNorm-fused GEMMs wherever supported, the fused launch of round 6 at every size without and
with its qkv tail, the defaults, two streams."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dino_amd  # noqa: E402
from dino_amd import DINOSeg, ViTConfig, procedural_state_dict  # noqa: E402
from dino_amd.weights import synthetic_frames  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
cfg = ViTConfig(n_blocks=3)
sd = procedural_state_dict(cfg)
m = DINOSeg(head="mlp", n_blocks=3, precision=prec, arch=cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m.to("cuda:0")
if prec in ("bf16", "fp16"):      # (option mlp_fused4 is read when the weights are packed: at the first forward)
    dino_amd.set_option("mlp_fused4", 1)
    m.set_resolution(64)
    m.forward_frames(torch.from_numpy(synthetic_frames(1, 64, seed=1)).cuda())
    torch.cuda.synchronize()
worst = 0.0
for c in range(cases):
    r = int(rng.choice([64, 96, 120, 168, 200, 248, 320, 400, 480]))
    B = int(rng.integers(1, 12))
    m.set_resolution(r)
    frames = torch.from_numpy(synthetic_frames(B, r, seed=1000 + c)).cuda()
    outs = {}
    single = prec in ("bf16", "fp16")
    if single:
        routes = (("separate", dict(mlp_fused=0, proj_fused=0, qkv_fused=0, streams=1, mlp_fused4=0)),
                  ("fused", dict(mlp_fused=2, proj_fused=0, qkv_fused=0, streams=1, mlp_fused4=0)),
                  ("fused+proj", dict(mlp_fused=2, proj_fused=1, qkv_fused=0, streams=1, mlp_fused4=0)),
                  ("fused+proj+qkv", dict(mlp_fused=2, proj_fused=1, qkv_fused=1, streams=1, mlp_fused4=0)),
                  ("one wave", dict(mlp_fused=2, proj_fused=1, qkv_fused=0, streams=1, mlp_fused4=1)),          # (round 6: mlp_fused4.hip)
                  ("two streams", dict(mlp_fused=2, proj_fused=1, qkv_fused=0, streams=2, split_min=2, mlp_fused4=0)),
                  ("two streams+qkv", dict(mlp_fused=2, proj_fused=1, qkv_fused=1, streams=2, split_min=2, mlp_fused4=0)))
        tol, flip_tol = (0.25, 0.02) if prec == "bf16" else (0.06, 0.01)
    else:
        routes = (("separate", dict(gemm_big=0, gemm_ln=0, streams=1, mlp_fused=0, qkv_fused3=1)),
                  ("persistent", dict(gemm_big=2, gemm_ln=0, streams=1, mlp_fused=0, qkv_fused3=1)),
                  ("ln-fused", dict(gemm_big=1, gemm_ln=2, streams=1, mlp_fused=0, qkv_fused3=1)),
                  ("fused3", dict(gemm_big=1, gemm_ln=1, streams=1, mlp_fused=2, qkv_fused3=0)),       # (round 6: mlp_fused3.hip at every size)
                  ("fused3+qkv", dict(gemm_big=1, gemm_ln=1, streams=1, mlp_fused=2, qkv_fused3=1)),   # ... with LayerNorm1 + qkv of the next block
                  ("fused+proj", dict(gemm_big=1, gemm_ln=1, streams=1, mlp_fused=1, qkv_fused3=1)),   # (the name the identity check below uses: the defaults)
                  ("two streams", dict(gemm_big=1, gemm_ln=1, streams=2, split_min=2, mlp_fused=1, qkv_fused3=1)))
        tol, flip_tol = (1.5e-3, 0.003) if prec == "bf16x3" else (3e-4, 0.001)
    for name, opts in routes:
        for k, v in opts.items():
            dino_amd.set_option(k, v)
        lp, am = m.forward_frames(frames)
        torch.cuda.synchronize()
        assert torch.isfinite(lp).all(), (name, r, B)
        outs[name] = (lp.clone(), am.clone())
    ref = outs["separate"][0]
    line = f"r={r:3d} B={B:2d} rows={B * ((r // 8) ** 2 + 1):6d}:"
    for name, (lp, am) in outs.items():
        if name == "separate":
            continue
        err = float((lp - ref).abs().max())
        flips = float((am != outs["separate"][1]).float().mean())
        worst = max(worst, err)
        line += f" {name} {err:.1e}/{100 * flips:.2f}%"
        assert err <= tol and flips <= flip_tol, (name, r, B, err, flips)
    assert torch.equal(outs["two streams"][0], outs["fused+proj"][0]), ("two streams differ", r, B)
    if single:
        assert not torch.equal(outs["one wave"][0], outs["fused+proj"][0]), ("option mlp_fused4 took no effect", r, B)
    if single:
        assert torch.equal(outs["two streams+qkv"][0], outs["fused+proj+qkv"][0]), ("two streams + qkv differ", r, B)
    print(line, flush=True)
for k, v in dict(mlp_fused=1, proj_fused=1, qkv_fused=0, streams=2, split_min=8, gemm_big=1, gemm_ln=1, mlp_fused4=0, qkv_fused3=1).items():
    dino_amd.set_option(k, v)
print(f"{cases} cases, worst |dlogp| between routes {worst:.3f}")
