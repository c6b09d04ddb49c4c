"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--only G1,G3,...]

What is the reference here:
  * ``src/vision_transformer.py`` imports cleanly (torch + numpy only).  Every ViT quantity below comes
    from the reference's own ``VisionTransformer`` loaded with the procedural weights
    (dino_amd/weights.py) via ``load_state_dict(strict=True)``.
  * ``src/pl_torch_modules.py`` (``DINOSeg``, ``MLP``) imports pytorch_lightning / albumentations /
    torchvision / comet at module top; none is installed and there is no network.  ``ref_dinoseg()``
    below pre-seeds ``sys.modules`` with minimal import stand-ins for those third-party names (SURVEY.md
    §8c) so that the reference's OWN ``DINOSeg.__init__/forward/predict/set_resolution/training_step``
    and ``MLP`` execute unmodified; ``get_dino`` (a network download, dt_utils.py:19-29) is pointed at a
    locally constructed ``vit_small(8)``.  The stand-ins contain no arithmetic of the path except
    albumentations' Normalize/ToTensorV2 formula (third-party, restated; parity unpinned there).

Only numbers are written (inputs, outputs); no reference source travels.
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from functools import partial

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/dt_segmentation"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from dino_amd.weights import (ViTConfig, procedural_state_dict, synthetic_frames, synthetic_labels)  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TINY = ViTConfig(embed_dim=128, num_heads=2, n_blocks=2)


def preprocess_np(frames_u8):
    """albumentations Normalize + ToTensorV2 (formula restated; used only to feed the reference ViT)."""
    mean = np.array((0.485, 0.456, 0.406), dtype=np.float32) * np.float32(255.0)
    inv = np.reciprocal(np.array((0.229, 0.224, 0.225), dtype=np.float32) * np.float32(255.0))
    x = (frames_u8.astype(np.float32) - mean) * inv
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))


def ref_vit(cfg: ViTConfig, sd):
    from src.vision_transformer import VisionTransformer
    m = VisionTransformer(patch_size=cfg.patch, embed_dim=cfg.embed_dim, depth=cfg.n_blocks, num_heads=cfg.num_heads,
                          mlp_ratio=cfg.mlp_ratio, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps),
                          num_classes=0)
    m.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("dino.")}, strict=True)
    return m.eval()


class TorchHead(nn.Module):
    """The head exactly as pl_torch_modules.py:108-124 composes it from torch.nn (Linear/ReLU/log_softmax)."""

    def __init__(self, cfg, sd):
        super().__init__()
        self.layer_1 = nn.Linear(cfg.embed_dim, 200)
        self.layer_2 = nn.Linear(200, 100)
        self.layer_3 = nn.Linear(100, cfg.n_classes)
        self.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("clf.")}, strict=True)

    def forward(self, x):
        x = torch.relu(self.layer_1(x))
        x = torch.relu(self.layer_2(x))
        return torch.log_softmax(self.layer_3(x), dim=1)


def ref_logp(vit, head, x):
    t = vit(x)[:, 1:]
    return head(t.reshape(-1, t.shape[-1]))


def margins(lp):
    top2 = torch.topk(lp, 2, dim=1).values
    return (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB")


# ----------------------------------------------------------------------------- G1 tiny ViT, intermediates
def g1():
    cfg, r = TINY, 64
    sd = procedural_state_dict(cfg)
    vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
    frames = synthetic_frames(2, r, seed=11)
    x = preprocess_np(frames)
    with torch.no_grad():
        tok = vit.prepare_tokens(x)
        outs, t = [], tok
        for blk in vit.blocks:
            t = blk(t)
            outs.append(t.numpy())
        final = vit.norm(t)
        lp = head(final[:, 1:].reshape(-1, cfg.embed_dim))
    save("g1_tiny_vit_r64", frames=frames, tokens=tok.numpy(), block1=outs[0], block2=outs[1], final=final.numpy(),
         logp=lp.numpy())


# ----------------------------------------------------------------------------- G2 pos-embed resample
def g2():
    cfg = ViTConfig(n_blocks=0)
    sd = procedural_state_dict(cfg)
    vit = ref_vit(cfg, sd)
    out = {}
    with torch.no_grad():
        for o in (30, 60, 120):
            dummy = torch.zeros(1, o * o + 1, cfg.embed_dim)
            pe = vit.interpolate_pos_encoding(dummy, o * 8, o * 8)
            out[f"o{o}"] = pe[0, :, :8].numpy().copy()
        dummy = torch.zeros(1, 28 * 28 + 1, cfg.embed_dim)
        out["o28"] = vit.interpolate_pos_encoding(dummy, 224, 224)[0, :, :8].numpy().copy()
    save("g2_pos_resample", **out)


# ----------------------------------------------------------------------------- G3 ViT-S/8 @480
def g3():
    r = 480
    frames = synthetic_frames(1, r, seed=0)
    x = preprocess_np(frames)
    for L in (1, 3, 12):
        cfg = ViTConfig(n_blocks=L)
        sd = procedural_state_dict(cfg)
        vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
        with torch.no_grad():
            lp = ref_logp(vit, head, x)
        save(f"g3_vits8_L{L}_r480", frame_seed=np.int64(0), logp=lp.numpy(), argmax=lp.argmax(1).numpy().astype(np.uint8),
             margin=margins(lp))


# ----------------------------------------------------------------------------- G4 ViT-S/8 L=3 @960
def g4():
    r, L = 960, 3
    frames = synthetic_frames(1, r, seed=4)
    cfg = ViTConfig(n_blocks=L)
    sd = procedural_state_dict(cfg)
    vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
    with torch.no_grad():
        lp = ref_logp(vit, head, preprocess_np(frames))
    rows = np.random.default_rng(44).choice(lp.shape[0], 256, replace=False).astype(np.int64)
    rows.sort()
    save("g4_vits8_L3_r960", frame_seed=np.int64(4), rows=rows, logp_rows=lp[rows].numpy(),
         argmax=lp.argmax(1).numpy().astype(np.uint8), margin=margins(lp))


# ----------------------------------------------------------------------------- G13 ViT-S/8 L=12 @960 (BASELINE configs[2] depth)
def g13():
    """The headline depth at 960x960 (14 401 tokens: the reference materialises a 6 x 14401^2 fp32 attention matrix per block,
    ~5 GB; twelve blocks run one after the other).  256 sampled rows of log-probabilities + the whole argmax map."""
    r, L = 960, 12
    frames = synthetic_frames(1, r, seed=4)
    cfg = ViTConfig(n_blocks=L)
    sd = procedural_state_dict(cfg)
    vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
    with torch.no_grad():
        lp = ref_logp(vit, head, preprocess_np(frames))
    rows = np.random.default_rng(45).choice(lp.shape[0], 256, replace=False).astype(np.int64)
    rows.sort()
    save("g4_vits8_L12_r960", frame_seed=np.int64(4), rows=rows, logp_rows=lp[rows].numpy(),
         argmax=lp.argmax(1).numpy().astype(np.uint8), margin=margins(lp))


# ----------------------------------------------------------------------------- G5 predict() maps
def g5():
    out = {}
    cfg = ViTConfig(n_blocks=3)
    sd = procedural_state_dict(cfg)
    vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
    for r in (240, 400, 480):
        frames = synthetic_frames(1, r, seed=50 + r, smooth=True)
        with torch.no_grad():
            lp = ref_logp(vit, head, preprocess_np(frames))
        o = r // 8
        low = torch.argmax(lp, dim=-1).numpy().reshape(o, o)
        k = 480 // o
        pred = np.kron(low, np.ones((k, k), dtype=int))      # pl_torch_modules.py:297-298
        out[f"frame_r{r}"] = frames[0]
        out[f"low_r{r}"] = low.astype(np.uint8)
        out[f"shape_r{r}"] = np.array(pred.shape, dtype=np.int64)
        out[f"margin_r{r}"] = margins(lp)
    save("g5_predict_L3", **out)


# ----------------------------------------------------------------------------- G6 fine-tune step
def _sample_idx(numel, k=64, seed=0):
    g = np.random.default_rng(seed)
    return np.sort(g.choice(numel, size=min(k, numel), replace=False)).astype(np.int64)


def g6():
    out = {}
    for tag, cfg, r, B in (("tiny_r64_B2", TINY, 64, 2), ("vits8_L3_r64_B2", ViTConfig(n_blocks=3), 64, 2)):
        sd = procedural_state_dict(cfg)
        vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
        vit.train(); head.train()
        frames = synthetic_frames(B, r, seed=61)
        labels = synthetic_labels(B, (r // 8) ** 2, cfg.n_classes, seed=62)
        x = preprocess_np(frames)
        y = torch.from_numpy(labels).reshape(-1).long()
        params = {("dino." + k): p for k, p in vit.named_parameters()}
        params.update({("clf." + k): p for k, p in head.named_parameters()})
        loss = torch.nn.functional.nll_loss(ref_logp(vit, head, x), y)     # pl_torch_modules.py:261-265
        loss.backward()
        out[f"{tag}/loss"] = np.float32(loss.item())
        for i, (k, p) in enumerate(params.items()):
            g = p.grad.detach().reshape(-1)
            idx = _sample_idx(g.numel(), 64, seed=i)
            out[f"{tag}/gnorm/{k}"] = np.float32(g.norm().item())
            out[f"{tag}/gidx/{k}"] = idx
            out[f"{tag}/gval/{k}"] = g[idx].numpy().copy()
        # 2 optimiser steps of Adam(lr=1e-3) (CLI default, run_experiment.py:135-136) and AdamW(lr=1e-6) (class default)
        for oname, ctor, lr in (("adam", torch.optim.Adam, 1e-3), ("adamw", torch.optim.AdamW, 1e-6)):
            vit2, head2 = ref_vit(cfg, sd), TorchHead(cfg, sd)
            vit2.train(); head2.train()
            p2 = {("dino." + k): p for k, p in vit2.named_parameters()}
            p2.update({("clf." + k): p for k, p in head2.named_parameters()})
            opt = ctor(list(p2.values()), lr=lr)
            losses = []
            for _ in range(2):
                opt.zero_grad()
                l2 = torch.nn.functional.nll_loss(ref_logp(vit2, head2, x), y)
                l2.backward()
                opt.step()
                losses.append(l2.item())
            out[f"{tag}/{oname}/losses"] = np.array(losses, dtype=np.float32)
            for i, (k, p) in enumerate(p2.items()):
                d = (p.detach() - torch.from_numpy(sd[k])).reshape(-1)
                idx = _sample_idx(d.numel(), 64, seed=i)
                out[f"{tag}/{oname}/delta/{k}"] = d[idx].numpy().copy()
    save("g6_finetune", **{k.replace("/", "|"): v for k, v in out.items()})


# ----------------------------------------------------------------------------- G7 ViT-B/8 L=12 @480
def g7():
    r = 480
    cfg = ViTConfig(embed_dim=768, num_heads=12, n_blocks=12)
    sd = procedural_state_dict(cfg)
    vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
    frames = synthetic_frames(1, r, seed=7)
    with torch.no_grad():
        lp = ref_logp(vit, head, preprocess_np(frames))
    rows = np.sort(np.random.default_rng(77).choice(lp.shape[0], 256, replace=False)).astype(np.int64)
    save("g7_vitb8_L12_r480", frame_seed=np.int64(7), rows=rows, logp_rows=lp[rows].numpy(),
         argmax=lp.argmax(1).numpy().astype(np.uint8), margin=margins(lp))


# ----------------------------------------------------------------------------- G9 the reference's own DINOSeg class
def ref_dinoseg():
    """Import the reference's pl_torch_modules with third-party import stand-ins (see module docstring)."""
    class _LM(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        def log(self, *a, **k):
            pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

    mod("pytorch_lightning", LightningModule=_LM, Trainer=_Any)
    mod("pytorch_lightning.callbacks")
    mod("pytorch_lightning.callbacks.early_stopping", EarlyStopping=_Any)
    mod("pytorch_lightning.callbacks.model_checkpoint", ModelCheckpoint=_Any)

    class Resize:
        def __init__(self, h, w):
            self.h, self.w = h, w

        def __call__(self, img):
            if img.shape[0] != self.h or img.shape[1] != self.w:
                raise RuntimeError("stand-in Resize only supports identity")
            return img

    class Normalize:
        def __init__(self, mean, std, max_pixel_value=255.0):
            self.mean = np.array(mean, dtype=np.float32) * np.float32(max_pixel_value)
            self.den = np.reciprocal(np.array(std, dtype=np.float32) * np.float32(max_pixel_value))

        def __call__(self, img):
            return (img.astype(np.float32) - self.mean) * self.den

    class ToTensorV2:
        def __call__(self, img):
            return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, image):
            for t in self.ts:
                image = t(image)
            return {"image": image}

    alb = mod("albumentations", Compose=Compose, Resize=Resize, Normalize=Normalize)
    for n in ("RandomResizedCrop", "ShiftScaleRotate", "HorizontalFlip", "ColorJitter", "GaussianBlur"):
        setattr(alb, n, _Any)
    mod("albumentations.pytorch", ToTensorV2=ToTensorV2)
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Resize=_Any, InterpolationMode=_Any, Compose=_Any, ToTensor=_Any,
                        Normalize=_Any, Grayscale=_Any)
    tv.models = mod("torchvision.models", resnet50=_Any)
    import src.pl_torch_modules as plm
    return plm


def g9():
    plm = ref_dinoseg()
    from src.vision_transformer import vit_small
    out = {}
    for L in (1, 3):
        cfg = ViTConfig(n_blocks=L)
        sd = procedural_state_dict(cfg)
        plm.get_dino = lambda patch_size=8, device="cpu": vit_small(patch_size=8, num_classes=0)
        m = plm.DINOSeg(data_path="", write_path="", head="mlp", n_blocks=L, n_classes=7, optimizer=torch.optim.Adam, lr=1e-3)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        m.eval()
        assert sorted(m.state_dict().keys()) == sorted(sd.keys())
        for r in (240, 480):
            m.set_resolution(r)
            frame = synthetic_frames(1, r, seed=90 + r + L)[0]
            pred = m.predict(frame)                                       # the reference's own predict()
            with torch.no_grad():
                lp = m(m.transforms(image=frame)["image"].unsqueeze(0))   # the reference's own forward()
            out[f"L{L}_r{r}_pred_shape"] = np.array(pred.shape, dtype=np.int64)
            out[f"L{L}_r{r}_pred_dtype"] = np.array(str(pred.dtype))
            o = r // 8
            k = 480 // o
            out[f"L{L}_r{r}_low"] = pred[::k, ::k].astype(np.uint8)
            assert np.array_equal(np.kron(pred[::k, ::k], np.ones((k, k), dtype=int)), pred)
            out[f"L{L}_r{r}_logp"] = lp.numpy()
            out[f"L{L}_r{r}_margin"] = margins(lp)
        m.set_resolution(400)
        out[f"L{L}_r400_pred_shape"] = np.array(m.predict(synthetic_frames(1, 400, seed=3)[0]).shape, dtype=np.int64)
        try:
            m.set_resolution(250)
        except ValueError as e:
            out["set_resolution_250_error"] = np.array(str(e))
        # training_step on B=2, r=64 (frozen / unfrozen gradient coverage + loss)
        m.train()
        frames = synthetic_frames(2, 64, seed=61)
        labels = synthetic_labels(2, 64, 7, seed=62)
        xb = torch.stack([m.transforms(image=f)["image"] for f in frames]) if False else preprocess_np(frames)
        for frozen in (True, False):
            m.freeze_bb() if frozen else m.unfreeze_bb()
            m.zero_grad()
            res = m.training_step((xb, torch.from_numpy(labels)), 0)
            res["loss"].backward()
            ng = sum(1 for p in m.parameters() if p.grad is not None and p.requires_grad)
            out[f"L{L}_train_{'frozen' if frozen else 'unfrozen'}_loss"] = np.float32(res["loss"].item())
            out[f"L{L}_train_{'frozen' if frozen else 'unfrozen'}_ngrad"] = np.int64(ng)
        opt = m.configure_optimizers()
        out[f"L{L}_optimizer"] = np.array(type(opt).__name__)
    save("g9_reference_dinoseg", **out)


# ----------------------------------------------------------------------------- G10 get_last_selfattention
def g10():
    out = {}
    cfg = TINY
    vit = ref_vit(cfg, procedural_state_dict(cfg))
    frames = synthetic_frames(1, 64, seed=101)
    with torch.no_grad():
        a = vit.get_last_selfattention(preprocess_np(frames))            # [1, H, N, N]
    out["tiny_r64_full"] = a.numpy()
    cfg = ViTConfig(n_blocks=3)
    vit = ref_vit(cfg, procedural_state_dict(cfg))
    frames = synthetic_frames(1, 96, seed=102)
    with torch.no_grad():
        a = vit.get_last_selfattention(preprocess_np(frames))
    out["vits8_L3_r96_cls_rows"] = a[0, :, 0, :].numpy().copy()           # what visualize_attention.py:50 uses
    out["vits8_L3_r96_row77"] = a[0, :, 77, :].numpy().copy()
    save("g10_last_selfattention", **out)


# ----------------------------------------------------------------------------- G14 get_intermediate_layers
def g14():
    """VisionTransformer.get_intermediate_layers(x, n) (vision_transformer.py:282-290): norm(x) after each of the last n blocks."""
    out = {}
    cfg = TINY
    vit = ref_vit(cfg, procedural_state_dict(cfg))
    frames = synthetic_frames(2, 64, seed=141)
    with torch.no_grad():
        for n in (1, 2, 5):                 # n beyond the depth returns every block's output
            ys = vit.get_intermediate_layers(preprocess_np(frames), n)
            out[f"tiny_r64_n{n}"] = np.stack([y.numpy() for y in ys])
    cfg = ViTConfig(n_blocks=3)
    vit = ref_vit(cfg, procedural_state_dict(cfg))
    frames2 = synthetic_frames(1, 96, seed=142)
    with torch.no_grad():
        ys = vit.get_intermediate_layers(preprocess_np(frames2), 2)
    out["vits8_L3_r96_n2"] = np.stack([y.numpy() for y in ys])[:, :, ::6, :].copy()      # every 6th token (CLS included)
    save("g14_intermediate_layers", frames_tiny=frames, frames_vits=frames2, **out)


def g11():
    """forward_mask / get_last_selfattention(x, cls_mask): the masked-CLS path of the last block."""
    out = {}
    for tag, cfg, r, nm in (("tiny_r64", TINY, 64, 3), ("vits8_L3_r96", ViTConfig(n_blocks=3), 96, 5)):
        vit = ref_vit(cfg, procedural_state_dict(cfg))
        frames = synthetic_frames(1, r, seed=111 + r)
        o = r // 8
        rng = np.random.default_rng(7 + r)
        masks = (rng.random((nm, o, o)) < 0.4).astype(np.float32)
        masks[0] = 1.0                                   # all keys
        masks[-1] = 0.0                                  # no key: uniform attention over zero logits
        with torch.no_grad():
            emb = vit.forward_mask(preprocess_np(frames), torch.from_numpy(masks))
            att = vit.get_last_selfattention(preprocess_np(frames), cls_mask=torch.from_numpy(masks))
        out[tag + "_masks"] = masks
        out[tag + "_emb"] = emb.numpy()
        out[tag + "_attn"] = att.numpy()
    save("g11_forward_mask", **out)


def g12():
    """Fine-tune step at the benchmark resolution ("r=480 B=1") and F.nll_loss's ignore_index rows.
    Separate file so that g6_finetune.npz stays byte-identical."""
    out = {}
    cases = (("vits8_L3_r480_B1", ViTConfig(n_blocks=3), 480, 1, 0.0), ("tiny_r64_B2_ignore", TINY, 64, 2, 0.25))
    for tag, cfg, r, B, ignore_frac in cases:
        sd = procedural_state_dict(cfg)
        vit, head = ref_vit(cfg, sd), TorchHead(cfg, sd)
        vit.train(); head.train()
        frames = synthetic_frames(B, r, seed=121)
        labels = synthetic_labels(B, (r // 8) ** 2, cfg.n_classes, seed=122).astype(np.int64)
        if ignore_frac > 0:
            drop = np.random.default_rng(123).random(labels.shape) < ignore_frac
            labels[drop] = -100                                      # F.nll_loss default ignore_index
            out[f"{tag}/labels"] = labels
        y = torch.from_numpy(labels).reshape(-1).long()
        params = {("dino." + k): p for k, p in vit.named_parameters()}
        params.update({("clf." + k): p for k, p in head.named_parameters()})
        loss = torch.nn.functional.nll_loss(ref_logp(vit, head, preprocess_np(frames)), y)     # pl_torch_modules.py:261-265
        loss.backward()
        out[f"{tag}/loss"] = np.float32(loss.item())
        for i, (k, p) in enumerate(params.items()):
            g = p.grad.detach().reshape(-1)
            idx = _sample_idx(g.numel(), 64, seed=i)
            out[f"{tag}/gnorm/{k}"] = np.float32(g.norm().item())
            out[f"{tag}/gidx/{k}"] = idx
            out[f"{tag}/gval/{k}"] = g[idx].numpy().copy()
    save("g12_finetune_r480_ignore", **{k.replace("/", "|"): v for k, v in out.items()})


ALL = {"G14": g14, "G13": g13, "G12": g12, "G11": g11, "G10": g10, "G1": g1, "G2": g2, "G3": g3, "G4": g4, "G5": g5, "G6": g6, "G7": g7, "G9": g9}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    names = [n for n in a.only.split(",") if n] or list(ALL)
    for n in names:
        print("==", n)
        ALL[n]()
