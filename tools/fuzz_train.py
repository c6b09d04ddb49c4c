"""GPU box: the fine-tune step on random (resolution, batch) shapes, both precisions: gradients with the weight-gradient side stream
(train_streams 2) against the one-stream walk (1e-4 of each tensor's largest entry; plain-store tensors are identical).  python tools/fuzz_train.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import dino_amd
from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
from dino_amd.weights import synthetic_frames, synthetic_labels
rng = np.random.default_rng(3)
for prec in ("bf16", "bf16x3"):
    cfg = ViTConfig(n_blocks=2)
    sd = procedural_state_dict(cfg)
    m = DINOSeg(head="mlp", n_blocks=2, precision=prec, arch=cfg, optimizer=torch.optim.Adam, lr=1e-3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.to("cuda:0"); m.unfreeze_bb()
    for c in range(10):
        r = int(rng.choice([64, 96, 120, 200, 320, 480])); B = int(rng.integers(1, 7))
        m.set_resolution(r)
        n = (r // 8) ** 2
        fr = torch.from_numpy(synthetic_frames(B, r, seed=300 + c)).cuda()
        lb = torch.from_numpy(synthetic_labels(B, n, cfg.n_classes, seed=400 + c)).cuda()
        gs = {}
        for ts in (1, 2):
            dino_amd.set_option("train_streams", ts)
            out = m.fused_training_step((fr, lb), 0)
            torch.cuda.synchronize()
            gs[ts] = {k: p.grad.clone() for k, p in m.named_parameters()}
            assert torch.isfinite(out["loss"])
        worst = 0.0
        for k in gs[1]:
            assert torch.isfinite(gs[2][k]).all(), k
            d = float((gs[1][k] - gs[2][k]).abs().max()) / (float(gs[1][k].abs().max()) + 1e-20)
            worst = max(worst, d)
        print(prec, "r", r, "B", B, "loss %.4f" % float(out["loss"]), "worst rel grad diff 1 vs 2 streams %.2e" % worst, flush=True)
        assert worst <= 1e-4
dino_amd.set_option("train_streams", 2)
print("ok")
