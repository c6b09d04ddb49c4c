"""Procedural, RNG-free synthetic weights for the DINOSeg path.

There is no network on the build or GPU boxes, so the pretrained DINO ViT weights the
reference downloads (``dt_segmentation/src/dt_utils.py:19-29``) and the Google-Drive
checkpoints (``README.md:9,28``) are unavailable.  Tests, golden fixtures and the benchmark
therefore use one deterministic recipe that depends only on integer hashing in numpy
(bit-identical on every machine, independent of any torch/numpy RNG implementation).

The tensor names and shapes are exactly the ``state_dict`` of the reference ``DINOSeg``
(SURVEY.md §5 "Checkpoint / resume"): ``dino.*`` for the truncated ViT
(``vision_transformer.py:161-196``) and ``clf.layer_{1,2,3}.*`` for the MLP head
(``pl_torch_modules.py:108-115``).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class ViTConfig:
    """Architecture of the truncated backbone + head (reference ctor kwargs that shape the path)."""
    embed_dim: int = 384        # vit_small, vision_transformer.py:300-304
    num_heads: int = 6
    n_blocks: int = 12          # DINOSeg truncates dino.blocks[:n_blocks], pl_torch_modules.py:177
    patch: int = 8
    mlp_ratio: int = 4
    n_classes: int = 7          # pl_torch_modules.py:146
    head: str = "mlp"           # 'mlp' | 'linear'  (pl_torch_modules.py:219-222)
    pos_grid: int = 28          # pos_embed is [1, 28*28+1, D] because img_size=[224], patch 8
    ln_eps: float = 1e-6        # partial(nn.LayerNorm, eps=1e-6), vision_transformer.py:303

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def hidden(self) -> int:
        return self.embed_dim * self.mlp_ratio


VIT_S8 = ViTConfig()
VIT_B8 = ViTConfig(embed_dim=768, num_heads=12)   # vit_base, vision_transformer.py:307-311


def tensor_shapes(cfg: ViTConfig) -> "OrderedDict[str, tuple]":
    """state_dict key -> shape, in the reference's registration order."""
    D, F, C, p = cfg.embed_dim, cfg.hidden, cfg.n_classes, cfg.patch
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["dino.cls_token"] = (1, 1, D)
    s["dino.pos_embed"] = (1, cfg.pos_grid * cfg.pos_grid + 1, D)
    s["dino.patch_embed.proj.weight"] = (D, 3, p, p)
    s["dino.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.n_blocks):
        b = f"dino.blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (F, D)
        s[b + "mlp.fc1.bias"] = (F,)
        s[b + "mlp.fc2.weight"] = (D, F)
        s[b + "mlp.fc2.bias"] = (D,)
    s["dino.norm.weight"] = (D,)
    s["dino.norm.bias"] = (D,)
    if cfg.head == "mlp":
        s["clf.layer_1.weight"] = (200, D)
        s["clf.layer_1.bias"] = (200,)
        s["clf.layer_2.weight"] = (100, 200)
        s["clf.layer_2.bias"] = (100,)
        s["clf.layer_3.weight"] = (C, 100)
        s["clf.layer_3.bias"] = (C,)
    else:
        s["clf.layer_1.weight"] = (C, D)
        s["clf.layer_1.bias"] = (C,)
    return s


def _hash_uniform(name: str, n: int, salt: int) -> np.ndarray:
    """n values in [-1, 1), from a 32-bit integer mix of (crc32(name), salt, index)."""
    seed = np.uint32(zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1 & 0xFFFFFFFF))
    x = np.arange(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        x = x * np.uint32(0x9E3779B1) + seed
        x ^= x >> np.uint32(16)
        x = x * np.uint32(0x85EBCA6B)
        x ^= x >> np.uint32(13)
        x = x * np.uint32(0xC2B2AE35)
        x ^= x >> np.uint32(16)
    # 24 high bits -> exactly representable fp32 in [-1, 1)
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23)) - np.float32(1.0)


def _scale_for(name: str, shape: tuple) -> tuple:
    """(offset, amplitude) of the uniform distribution for tensor `name`."""
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "dino.norm.weight":
        return 1.0, 0.25
    if "norm" in name and name.endswith(".bias"):
        return 0.0, 0.10
    if name.endswith(".bias"):
        return 0.0, 0.05
    if name == "dino.cls_token":
        return 0.0, 0.30
    if name == "dino.pos_embed":
        return 0.0, 0.20
    if name == "dino.patch_embed.proj.weight":
        return 0.0, 0.06
    if name.endswith("attn.qkv.weight"):
        return 0.0, 0.14          # std 0.08 -> q,k std ~1.5 -> softmax scores std ~2
    if name.endswith("attn.proj.weight"):
        return 0.0, 0.05
    if name.endswith("mlp.fc1.weight"):
        return 0.0, 0.09
    if name.endswith("mlp.fc2.weight"):
        return 0.0, 0.04
    if name == "clf.layer_1.weight":
        return 0.0, 0.14 if len(shape) == 2 and shape[0] == 200 else 0.3
    if name == "clf.layer_2.weight":
        return 0.0, 0.20
    if name == "clf.layer_3.weight":
        return 0.0, 0.60
    return 0.0, 0.05


def procedural_state_dict(cfg: ViTConfig = VIT_S8, salt: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 weights for every tensor of the path, as numpy arrays."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in tensor_shapes(cfg).items():
        n = int(np.prod(shape))
        off, amp = _scale_for(name, shape)
        if cfg.embed_dim > 384 and name.endswith(".weight") and len(shape) == 2 and "norm" not in name:
            # keep activation / logit magnitudes of the ViT-S recipe when the fan-in grows with embed_dim (ViT-B)
            amp *= float(np.sqrt(384.0 / cfg.embed_dim))
        w = _hash_uniform(name, n, salt) * np.float32(amp) + np.float32(off)
        if name == "dino.pos_embed":
            # low-frequency structure on top of the hash so the bicubic resample is exercised
            g = cfg.pos_grid
            yy, xx = np.meshgrid(np.arange(g, dtype=np.float32), np.arange(g, dtype=np.float32), indexing="ij")
            d = np.arange(shape[2], dtype=np.float32)
            wave = 0.25 * np.sin(0.21 * yy[..., None] + 0.013 * d) * np.cos(0.17 * xx[..., None] - 0.007 * d)
            w = w.reshape(shape)
            w[0, 1:, :] += wave.reshape(g * g, shape[2]).astype(np.float32)
        w = w.reshape(shape)
        if name in ("clf.layer_2.weight", "clf.layer_3.weight") or (name == "clf.layer_1.weight" and cfg.head == "linear"):
            # zero row mean: a constant (post-ReLU / common-mode) input favours no unit or class by construction
            w = w - w.mean(axis=1, keepdims=True, dtype=np.float32)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def synthetic_frames(B: int, r: int, seed: int = 0, smooth: bool = False) -> np.ndarray:
    """uint8 [B, r, r, 3] frames already at r x r (resize = identity)."""
    if not smooth:
        return np.random.default_rng(seed).integers(0, 256, (B, r, r, 3), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(r, dtype=np.float32), np.arange(r, dtype=np.float32), indexing="ij")
    frames = np.empty((B, r, r, 3), dtype=np.uint8)
    for b in range(B):
        for c in range(3):
            f = (np.sin(0.031 * (c + 1) * xx + 0.7 * b + seed) * np.cos(0.017 * (3 - c) * yy - 0.3 * c)
                 + 0.35 * np.sin(0.11 * (xx + yy) + c))
            frames[b, :, :, c] = np.clip(127.5 + 94.0 * f, 0, 255).astype(np.uint8)
    return frames


def synthetic_labels(B: int, n: int, n_classes: int = 7, seed: int = 1) -> np.ndarray:
    """int64 [B, n] patch labels for the fine-tune step."""
    return np.random.default_rng(seed).integers(0, n_classes, (B, n)).astype(np.int64)
