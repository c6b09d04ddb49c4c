"""CPU tests: the oracle (oracle/dinoseg_oracle.py) against the golden vectors captured from the reference
(oracle/gen_golden.py -> tests/golden/*.npz).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from dino_amd.weights import ViTConfig, procedural_state_dict, synthetic_frames, synthetic_labels
from oracle import dinoseg_oracle as O

TINY = ViTConfig(embed_dim=128, num_heads=2, n_blocks=2)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_g1_tiny_vit_intermediates(golden_dir):
    g = load(golden_dir, "g1_tiny_vit_r64")
    W = O.to_torch(procedural_state_dict(TINY))
    taps = {}
    with torch.no_grad():
        x = O.preprocess(g["frames"])
        final = O.vit_forward(x, W, TINY.num_heads, taps=taps)
        lp = O.head_forward(final[:, 1:].reshape(-1, TINY.embed_dim), W)
    for key, val in (("tokens", taps["tokens"]), ("block1", taps["block0"]), ("block2", taps["block1"]), ("final", final),
                     ("logp", lp)):
        assert float((val - torch.from_numpy(g[key])).abs().max()) <= 2e-5, key


def test_g2_pos_resample(golden_dir):
    g = load(golden_dir, "g2_pos_resample")
    pe = torch.from_numpy(procedural_state_dict(ViTConfig(n_blocks=0))["dino.pos_embed"])
    for o in (28, 30, 60, 120):
        mine = O.resample_pos_embed(pe, o)[0, :, :8]
        assert mine.shape == g[f"o{o}"].shape
        assert float((mine - torch.from_numpy(g[f"o{o}"])).abs().max()) <= 2e-6, o


@pytest.mark.parametrize("L", [1, 3, 12])
def test_g3_vits8_480(golden_dir, L):
    g = load(golden_dir, f"g3_vits8_L{L}_r480")
    cfg = ViTConfig(n_blocks=L)
    W = O.to_torch(procedural_state_dict(cfg))
    with torch.no_grad():
        lp = O.dinoseg_forward(O.preprocess(synthetic_frames(1, 480, seed=int(g["frame_seed"]))), W, cfg.num_heads)
    assert float((lp - torch.from_numpy(g["logp"])).abs().max()) <= 5e-5
    assert np.array_equal(lp.argmax(1).numpy(), g["argmax"].astype(np.int64))


def test_g5_predict_maps(golden_dir):
    g = load(golden_dir, "g5_predict_L3")
    cfg = ViTConfig(n_blocks=3)
    W = O.to_torch(procedural_state_dict(cfg))
    for r in (240, 400):
        pred = O.predict(g[f"frame_r{r}"], W, cfg.num_heads, r)
        assert pred.dtype == np.int64 and tuple(pred.shape) == tuple(g[f"shape_r{r}"])
        k = 480 // (r // 8)
        low = pred[::k, ::k]
        same = low == g[f"low_r{r}"].astype(np.int64)
        # fp32 CPU summation order differs between the oracle and the reference; only near-ties may move
        assert np.all(same | (g[f"margin_r{r}"].reshape(low.shape) < 1e-4))
    assert tuple(g["shape_r400"]) == (450, 450)      # the reference's r=400 quirk: 480 // 50 = 9
    with pytest.raises(ValueError, match="Resolution should be a multiple of 8."):
        O.predict(np.zeros((250, 250, 3), np.uint8), W, cfg.num_heads, 250)


def test_g9_reference_dinoseg_class(golden_dir):
    """Outputs of the reference's own DINOSeg.forward / predict / training_step (run under import stand-ins)."""
    g = load(golden_dir, "g9_reference_dinoseg")
    assert str(g["set_resolution_250_error"]) == "Resolution should be a multiple of 8."
    for L in (1, 3):
        cfg = ViTConfig(n_blocks=L)
        sd = procedural_state_dict(cfg)
        W = O.to_torch(sd)
        r = 240
        frame = synthetic_frames(1, r, seed=90 + r + L)[0]
        with torch.no_grad():
            lp = O.dinoseg_forward(O.preprocess(frame[None]), W, cfg.num_heads)
        assert float((lp - torch.from_numpy(g[f"L{L}_r{r}_logp"])).abs().max()) <= 5e-5
        pred = O.predict(frame, W, cfg.num_heads, r)
        assert tuple(pred.shape) == tuple(g[f"L{L}_r{r}_pred_shape"]) == (480, 480)
        assert str(g[f"L{L}_r{r}_pred_dtype"]) == str(pred.dtype) == "int64"
        k = 480 // (r // 8)
        same = pred[::k, ::k] == g[f"L{L}_r{r}_low"].astype(np.int64)
        assert np.all(same | (g[f"L{L}_r{r}_margin"].reshape(same.shape) < 1e-4))
        assert tuple(g[f"L{L}_r400_pred_shape"]) == (450, 450)
        # training_step loss (mean NLL over B*n patches) and gradient coverage: 6 head tensors frozen, all unfrozen
        Wg = O.to_torch(sd, requires_grad=True)
        frames = synthetic_frames(2, 64, seed=61)
        labels = torch.from_numpy(synthetic_labels(2, 64, 7, seed=62))
        loss = O.nll_loss(O.dinoseg_forward(O.preprocess(frames), Wg, cfg.num_heads), labels)
        assert abs(float(loss) - float(g[f"L{L}_train_unfrozen_loss"])) <= 2e-5
        assert abs(float(loss) - float(g[f"L{L}_train_frozen_loss"])) <= 2e-5
        assert int(g[f"L{L}_train_frozen_ngrad"]) == 6
        assert int(g[f"L{L}_train_unfrozen_ngrad"]) == len(sd)
        assert str(g[f"L{L}_optimizer"]) == "Adam"


def test_g6_finetune_gradients(golden_dir):
    g = load(golden_dir, "g6_finetune")
    for tag, cfg in (("tiny_r64_B2", TINY), ("vits8_L3_r64_B2", ViTConfig(n_blocks=3))):
        sd = procedural_state_dict(cfg)
        W = O.to_torch(sd, requires_grad=True)
        frames = synthetic_frames(2, 64, seed=61)
        labels = torch.from_numpy(synthetic_labels(2, 64, cfg.n_classes, seed=62))
        loss = O.nll_loss(O.dinoseg_forward(O.preprocess(frames), W, cfg.num_heads), labels)
        loss.backward()
        assert abs(float(loss) - float(g[f"{tag}|loss"])) <= 2e-5
        for k, p in W.items():
            gv = p.grad.reshape(-1)
            gn = float(g[f"{tag}|gnorm|{k}"])
            assert abs(float(gv.norm()) - gn) <= 1e-4 * gn + 1e-7, k
            idx = torch.from_numpy(g[f"{tag}|gidx|{k}"])
            assert float((gv[idx] - torch.from_numpy(g[f"{tag}|gval|{k}"])).abs().max()) <= 2e-4 * gn + 1e-7, k


def test_g12_finetune_r480_and_ignore_index(golden_dir):
    """G12: the fine-tune step at the benchmark resolution (B=1, 3601 tokens) and with F.nll_loss's ignore_index rows."""
    g = load(golden_dir, "g12_finetune_r480_ignore")
    for tag, cfg, r, B in (("vits8_L3_r480_B1", ViTConfig(n_blocks=3), 480, 1), ("tiny_r64_B2_ignore", TINY, 64, 2)):
        sd = procedural_state_dict(cfg)
        W = O.to_torch(sd, requires_grad=True)
        frames = synthetic_frames(B, r, seed=121)
        key = f"{tag}|labels"
        labels = torch.from_numpy(g[key] if key in g.files else synthetic_labels(B, (r // 8) ** 2, cfg.n_classes, seed=122))
        if key in g.files:
            assert int((labels == -100).sum()) > 10
        loss = O.nll_loss(O.dinoseg_forward(O.preprocess(frames), W, cfg.num_heads), labels)
        loss.backward()
        assert abs(float(loss) - float(g[f"{tag}|loss"])) <= 2e-5
        for k, p in W.items():
            gv = p.grad.reshape(-1)
            gn = float(g[f"{tag}|gnorm|{k}"])
            assert abs(float(gv.norm()) - gn) <= 1e-4 * gn + 1e-7, k
            idx = torch.from_numpy(g[f"{tag}|gidx|{k}"])
            assert float((gv[idx] - torch.from_numpy(g[f"{tag}|gval|{k}"])).abs().max()) <= 2e-4 * gn + 1e-7, k
    with pytest.raises(IndexError):
        O.nll_loss(torch.zeros(4, 7), torch.tensor([0, 7, 1, 2]))


def test_g10_last_selfattention(golden_dir):
    g = load(golden_dir, "g10_last_selfattention")
    W = O.to_torch(procedural_state_dict(TINY))
    with torch.no_grad():
        a = O.last_selfattention(O.preprocess(synthetic_frames(1, 64, seed=101)), W, TINY.num_heads)
    assert tuple(a.shape) == tuple(g["tiny_r64_full"].shape) == (1, 2, 65, 65)
    assert float((a - torch.from_numpy(g["tiny_r64_full"])).abs().max()) <= 1e-6
    cfg = ViTConfig(n_blocks=3)
    W = O.to_torch(procedural_state_dict(cfg))
    with torch.no_grad():
        a = O.last_selfattention(O.preprocess(synthetic_frames(1, 96, seed=102)), W, cfg.num_heads)
    assert float((a[0, :, 0] - torch.from_numpy(g["vits8_L3_r96_cls_rows"])).abs().max()) <= 5e-6
    assert float((a[0, :, 77] - torch.from_numpy(g["vits8_L3_r96_row77"])).abs().max()) <= 5e-6


def test_g14_intermediate_layers(golden_dir):
    """VisionTransformer.get_intermediate_layers(x, n) of the reference (vision_transformer.py:282-290) vs the oracle."""
    g = load(golden_dir, "g14_intermediate_layers")
    W = O.to_torch(procedural_state_dict(TINY))
    x = O.preprocess(g["frames_tiny"])
    with torch.no_grad():
        for n, want in ((1, 1), (2, 2), (5, 2)):
            ys = O.intermediate_layers(x, W, TINY.num_heads, n)
            ref = torch.from_numpy(g[f"tiny_r64_n{n}"])
            assert len(ys) == want == ref.shape[0]
            assert float((torch.stack(ys) - ref).abs().max()) <= 2e-5
        assert O.intermediate_layers(x, W, TINY.num_heads, 0) == []
        cfg = ViTConfig(n_blocks=3)
        ys = O.intermediate_layers(O.preprocess(g["frames_vits"]), O.to_torch(procedural_state_dict(cfg)), cfg.num_heads, 2)
        assert float((torch.stack(ys)[:, :, ::6] - torch.from_numpy(g["vits8_L3_r96_n2"])).abs().max()) <= 5e-5


def test_metrics_from_confusion_match_sklearn():
    from sklearn.metrics import balanced_accuracy_score, f1_score, jaccard_score
    from dino_amd.dinoseg import metrics_from_confusion
    rng = np.random.default_rng(5)
    for C, skew in ((7, False), (7, True), (3, False)):
        gt = rng.integers(0, C if not skew else C - 2, 5000)          # skew: two classes never occur in gt
        pred = np.where(rng.random(5000) < 0.7, gt, rng.integers(0, C if not skew else C - 1, 5000))
        cm = np.zeros((C, C))
        np.add.at(cm, (gt, pred), 1)
        m = metrics_from_confusion(cm, "val")
        assert abs(m["val_acc"] - balanced_accuracy_score(gt, pred)) <= 1e-12
        assert abs(m["val_F1"] - f1_score(gt, pred, average="macro")) <= 1e-12
        assert abs(m["val_iou"] - jaccard_score(gt, pred, average="macro")) <= 1e-12


def test_g11_forward_mask(golden_dir):
    """forward_mask / get_last_selfattention(x, cls_mask) of the reference ViT (vision_transformer.py:250-280) vs the oracle."""
    from dino_amd.weights import ViTConfig
    g = np.load(os.path.join(golden_dir, "g11_forward_mask.npz"))
    for tag, cfg, r in (("tiny_r64", ViTConfig(embed_dim=128, num_heads=2, n_blocks=2), 64), ("vits8_L3_r96", ViTConfig(n_blocks=3), 96)):
        W = O.to_torch(procedural_state_dict(cfg))
        x = O.preprocess(synthetic_frames(1, r, seed=111 + r))
        m = torch.from_numpy(g[tag + "_masks"])
        emb = O.forward_mask(x, W, cfg.num_heads, m)
        att = O.forward_mask(x, W, cfg.num_heads, m, return_attention=True)
        assert float((emb - torch.from_numpy(g[tag + "_emb"])).abs().max()) <= 2e-5
        assert float((att - torch.from_numpy(g[tag + "_attn"])).abs().max()) <= 1e-5
