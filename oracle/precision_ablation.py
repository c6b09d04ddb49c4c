#!/usr/bin/env python3
"""Which operands of the six products per block need bf16 hi+lo planes?  -- TEST INFRASTRUCTURE ONLY (CPU).

The HIP path stores every GEMM / attention operand as bf16 planes: 1 plane = plain bf16 (8 mantissa bits),
2 planes = hi + lo split (~16 bits).  A product with planes (pa, pb) costs pa*pb MFMAs minus the dropped lo*lo term:
(1,1) = 1, (2,1) = (1,2) = 2, (2,2) = 3.  The operand rounding is the whole numerical difference between the modes
(accumulation, LayerNorm, softmax statistics and the residual stream are fp32 in every mode), so it can be studied on
the CPU: this script re-runs the oracle forward with each operand rounded to its plane count and measures
max |dlogp| and argmax flips against the golden log-probabilities captured from the reference
(tests/golden/g3_vits8_L{L}_r480.npz, g4 @960, g7 ViT-B).

Operand slots (12 per block), in kernel order:
  qkv.A  LN1 output         qkv.W  attn.qkv.weight
  s.Q    stored Q           s.K    stored K            (QK^T)
  pv.P   probabilities      pv.V   stored V            (P V)
  proj.A stored ctx         proj.W attn.proj.weight
  fc1.A  LN2 output         fc1.W  mlp.fc1.weight
  fc2.A  stored GELU output fc2.W  mlp.fc2.weight
The patch embedding and the classifier head always run split (they are < 0.4 % of the FLOPs).

    python oracle/precision_ablation.py [--golden g3_vits8_L12_r480] [--greedy] [--config name=planes,...]

Writes a markdown table to stdout (committed as profiles/r02_precision_ablation.md).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dino_amd.weights import VIT_B8, ViTConfig, procedural_state_dict, synthetic_frames  # noqa: E402
from oracle import dinoseg_oracle as O  # noqa: E402

SLOTS = ["qkv.A", "qkv.W", "s.Q", "s.K", "pv.P", "pv.V", "proj.A", "proj.W", "fc1.A", "fc1.W", "fc2.A", "fc2.W"]


def quant_fp16x2(x):
    hi = x.to(torch.float16).to(torch.float32)
    return hi + (x - hi).to(torch.float16).to(torch.float32)


def rnd(x, planes):
    """planes: 1 = bf16, 2 = bf16 hi+lo, 3 = exact fp32, 11 = fp16 (one plane), 12 = fp16 hi+lo"""
    if planes == 3:
        return x
    if planes == 11:
        return O.quant_fp16(x)
    if planes == 12:
        return quant_fp16x2(x)
    return O.quant_bf16(x) if planes == 1 else O.quant_bf16x2(x)


def forward(x, W, H, cfg, eps=1e-6):
    """oracle forward with per-slot operand rounding; cfg: slot -> planes (1, 2; 3 = exact fp32)"""
    q2 = O.quant_bf16x2
    t = O.prepare_tokens(x, W, 8, q2)
    L = O.count_blocks(W)
    for i in range(L):
        pre = f"dino.blocks.{i}."
        B, N, D = t.shape
        dh = D // H
        a = O.layer_norm(t, W[pre + "norm1.weight"], W[pre + "norm1.bias"], eps)
        qkv = rnd(a, cfg["qkv.A"]) @ rnd(W[pre + "attn.qkv.weight"], cfg["qkv.W"]).t() + W[pre + "attn.qkv.bias"]
        qkv = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        # the kernel stores Q pre-scaled by dh^-0.5 * log2(e); a power-of-two-free scale moves the rounding points slightly
        # but not the error magnitude
        Q = rnd(qkv[0] * (dh ** -0.5), cfg["s.Q"])
        K = rnd(qkv[1], cfg["s.K"])
        V = rnd(qkv[2], cfg["pv.V"])
        ctx = torch.empty(B, H, N, dh)
        for b in range(B):
            for h in range(H):          # one head at a time: 52 MB of scores instead of 311 MB
                s = Q[b, h] @ K[b, h].t()
                s = s - s.amax(dim=-1, keepdim=True)
                e = torch.exp(s)
                # the kernel rounds the UNNORMALISED probabilities (<= 1 relative to the running reference) and divides
                # by the fp32 row sum afterwards
                ctx[b, h] = (rnd(e, cfg["pv.P"]) @ V[b, h]) / e.sum(dim=-1, keepdim=True)
        ctx = ctx.transpose(1, 2).reshape(B, N, D)
        t = t + rnd(ctx, cfg["proj.A"]) @ rnd(W[pre + "attn.proj.weight"], cfg["proj.W"]).t() + W[pre + "attn.proj.bias"]
        a = O.layer_norm(t, W[pre + "norm2.weight"], W[pre + "norm2.bias"], eps)
        hdn = O.gelu_erf(rnd(a, cfg["fc1.A"]) @ rnd(W[pre + "mlp.fc1.weight"], cfg["fc1.W"]).t() + W[pre + "mlp.fc1.bias"])
        t = t + rnd(hdn, cfg["fc2.A"]) @ rnd(W[pre + "mlp.fc2.weight"], cfg["fc2.W"]).t() + W[pre + "mlp.fc2.bias"]
    t = O.layer_norm(t, W["dino.norm.weight"], W["dino.norm.bias"], eps)[:, 1:]
    return O.head_forward(t.reshape(-1, t.shape[-1]), W, q2)


def mfma_cost(cfg):
    """MFMA count per block relative to plain bf16 (FLOP-weighted): qkv 3.186, QK 9.959, PV 9.959, proj 1.062, fc1/fc2 4.248"""
    w = {"qkv": 3.186, "s": 9.959, "pv": 9.959, "proj": 1.062, "fc1": 4.248, "fc2": 4.248}
    pairs = {"qkv": ("qkv.A", "qkv.W"), "s": ("s.Q", "s.K"), "pv": ("pv.P", "pv.V"), "proj": ("proj.A", "proj.W"),
             "fc1": ("fc1.A", "fc1.W"), "fc2": ("fc2.A", "fc2.W")}
    tot = 0.0
    for k, (a, b) in pairs.items():
        pa, pb = (1 if cfg[a] in (1, 11) else 2), (1 if cfg[b] in (1, 11) else 2)
        tot += w[k] * (pa * pb - (1 if pa == 2 and pb == 2 else 0))
    return tot / sum(w.values())


def load_case(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    vitb = "vitb" in name
    L = int(name.split("_L")[1].split("_")[0])
    r = int(name.split("_r")[1])
    base = VIT_B8 if vitb else ViTConfig()
    cfg = ViTConfig(embed_dim=base.embed_dim, num_heads=base.num_heads, n_blocks=L)
    W = O.to_torch(procedural_state_dict(cfg))
    x = O.preprocess(synthetic_frames(1, r, seed=int(g["frame_seed"])))
    if "logp" in g.files:
        rows, ref = None, torch.from_numpy(g["logp"])
    else:
        rows, ref = torch.from_numpy(g["rows"]), torch.from_numpy(g["logp_rows"])
    return cfg, W, x, rows, ref, torch.from_numpy(g["argmax"].astype(np.int64)), torch.from_numpy(g["margin"])


def evaluate(case, pcfg):
    cfg, W, x, rows, ref, ref_arg, margin = case
    with torch.no_grad():
        lp = forward(x, W, cfg.num_heads, pcfg)
    err = float(((lp if rows is None else lp[rows]) - ref).abs().max())
    arg = lp.argmax(1)
    flips = int((arg != ref_arg).sum())
    return err, flips


def fmt(pcfg):
    return " ".join(f"{s}={pcfg[s]}" for s in SLOTS)


NAMED = {
    "bf16": {s: 1 for s in SLOTS},
    "bf16x3": {s: 2 for s in SLOTS},
    "fp16": {s: 11 for s in SLOTS},
    "fp16x3": {s: 12 for s in SLOTS},
    # judge's starting point: QK^T and the LN-fed GEMMs split, P.V / fc2 / proj single, weight-side lo plane dropped
    "verdict_start": {"qkv.A": 2, "qkv.W": 1, "s.Q": 2, "s.K": 2, "pv.P": 1, "pv.V": 1, "proj.A": 1, "proj.W": 1,
                      "fc1.A": 2, "fc1.W": 1, "fc2.A": 1, "fc2.W": 1},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--golden", default="g3_vits8_L12_r480")
    ap.add_argument("--greedy", action="store_true", help="from all-split, drop lo planes one at a time while the bar holds")
    ap.add_argument("--single", action="store_true", help="all-split with exactly one slot single, and all-single with one split")
    ap.add_argument("--config", action="append", default=[], help="name or slot=planes,... (unlisted slots = 2)")
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    case = load_case(a.golden)
    print(f"# precision ablation on {a.golden} (CPU emulation of operand rounding; oracle/precision_ablation.py)\n")
    print("| config | MFMAs / bf16 MFMAs | max abs dlogp | argmax flips |")
    print("|---|---|---|---|")

    def run(name, pcfg):
        t0 = time.time()
        err, flips = evaluate(case, pcfg)
        print(f"| {name} | {mfma_cost(pcfg):.3f} | {err:.3e} | {flips} |  <!-- {time.time() - t0:.0f}s -->", flush=True)
        return err, flips

    for c in a.config:
        if c in NAMED:
            run(c, NAMED[c])
        else:
            pc = {s: 2 for s in SLOTS}
            for kv in c.split(","):
                k, v = kv.split("=")
                pc[k] = int(v)
            run(c, pc)
    if a.single:
        run("bf16x3 (all split)", NAMED["bf16x3"])
        for s in SLOTS:
            pc = dict(NAMED["bf16x3"])
            pc[s] = 1
            run(f"all split, {s} single", pc)
        run("bf16 (all single)", NAMED["bf16"])
        for s in SLOTS:
            pc = dict(NAMED["bf16"])
            pc[s] = 2
            run(f"all single, {s} split", pc)
    if a.greedy:
        pc = dict(NAMED["bf16x3"])
        err, flips = run("start: all split", pc)
        # cheapest-first: try dropping the lo plane of the slots with the largest MFMA saving first
        order = ["pv.P", "pv.V", "s.Q", "s.K", "fc2.A", "fc2.W", "fc1.A", "fc1.W", "qkv.A", "qkv.W", "proj.A", "proj.W"]
        for s in order:
            trial = dict(pc)
            trial[s] = 1
            e, f = run(f"drop {s}", trial)
            if e <= a.tol * 0.6 and f == 0:      # keep 40 % headroom for the other goldens / accumulation-order noise
                pc = trial
        run("greedy result: " + fmt(pc), pc)


if __name__ == "__main__":
    main()
