"""CPU fp32 restatement of the reference DINOSeg hot path -- TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: an independent, functional, weight-dict-driven restatement (plain
PyTorch CPU fp32 ops, own structure, nothing copied) of what the reference computes on the
path SURVEY.md §8a lists.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product (``dino_amd``) never does.

Pinning: the reference has no tests or golden vectors of its own.  The oracle
is pinned against outputs of the reference itself: ``oracle/gen_golden.py`` imports the
reference's ``src/vision_transformer.py`` in the build container, loads the procedural
weights, and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file
against those fixtures.

Reference lines restated (all under /root/reference/dt_segmentation/src/):
  preprocess            pl_torch_modules.py:33-41, :291     (albumentations Normalize + ToTensorV2)
  patch_embed           vision_transformer.py:143-158
  resample_pos_embed    vision_transformer.py:202-222       (torch bicubic, scale_factor rule)
  prepare_tokens        vision_transformer.py:224-235
  layer_norm            vision_transformer.py:303 (eps 1e-6), :114,:118,:183
  attention             vision_transformer.py:68-107
  mlp                   vision_transformer.py:49-65         (exact erf GELU)
  block                 vision_transformer.py:110-140
  vit_forward           vision_transformer.py:237-248
  last_selfattention    vision_transformer.py:273-280
  head_forward          pl_torch_modules.py:108-138
  dinoseg_forward       pl_torch_modules.py:239-256
  predict               pl_torch_modules.py:270-300
  nll_loss / train step pl_torch_modules.py:258-268
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch

Tensor = torch.Tensor
_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def _ident(x: Tensor) -> Tensor:
    return x


# --------------------------------------------------------------------------- preprocessing
def preprocess(frames_u8: np.ndarray) -> Tensor:
    """uint8 [B,r,r,3] (already r x r: resize is the identity) -> fp32 [B,3,r,r].

    albumentations.Normalize computes (img - mean*255) * (1/(std*255)) in fp32
    (pl_torch_modules.py:37); ToTensorV2 moves HWC -> CHW (:38).
    """
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).to(torch.float32)
    mean = torch.tensor(_IMAGENET_MEAN, dtype=torch.float32) * 255.0
    inv = 1.0 / (torch.tensor(_IMAGENET_STD, dtype=torch.float32) * 255.0)
    x = (x - mean) * inv
    return x.permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------- pos-embed resample
def _cubic_weights(t: Tensor, A: float = -0.75):
    """Keys cubic convolution coefficients for taps at -1, 0, +1, +2 (A = -0.75 as ATen)."""
    def near(x):   # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def far(x):    # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)


def resample_pos_embed(pos_embed: Tensor, o: int) -> Tensor:
    """[1, g*g+1, D] stored pos-embed -> [1, o*o+1, D] for an o x o token grid.

    The reference calls F.interpolate(mode='bicubic', scale_factor=(o+0.1)/g) with
    align_corners=False (vision_transformer.py:212-219); ATen then maps destination index d
    to source coordinate (d + 0.5) / scale_factor - 0.5 -- using the *given* scale factor
    (g/(o+0.1)), not the size ratio g/o -- and clamps the 4 taps to the border.
    """
    g2 = pos_embed.shape[1] - 1
    g = int(round(math.sqrt(g2)))
    D = pos_embed.shape[2]
    if o == g:
        return pos_embed
    cls_pos = pos_embed[:, :1]
    grid = pos_embed[0, 1:].reshape(g, g, D)
    scale = float(g) / (float(o) + 0.1)
    dst = torch.arange(o, dtype=torch.float32)
    src = (dst + 0.5) * scale - 0.5
    base = torch.floor(src)
    t = src - base
    base = base.to(torch.int64)
    w = torch.stack(_cubic_weights(t), dim=1)                                   # [o,4]
    idx = torch.stack([(base + k).clamp(0, g - 1) for k in (-1, 0, 1, 2)], 1)   # [o,4]
    # separable: rows (y) then columns (x); accumulate in fp32 like ATen
    rows = (grid[idx] * w[:, :, None, None]).sum(1)                             # [o,g,D]
    out = (rows[:, idx] * w[None, :, :, None]).sum(2)                           # [o,o,D]
    return torch.cat([cls_pos, out.reshape(1, o * o, D)], dim=1)


# --------------------------------------------------------------------------- ViT pieces
def layer_norm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * g + b


def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], q: Callable = _ident) -> Tensor:
    y = q(x) @ q(w).t()
    return y if b is None else y + b


def patch_embed(x: Tensor, W: Dict[str, Tensor], p: int, q: Callable = _ident) -> Tensor:
    """[B,3,r,r] -> [B,n,D]; the k index of the GEMM is c*p*p + ky*p + kx, rows are py*o+px."""
    B, C, H, Wd = x.shape
    oy, ox = H // p, Wd // p
    cols = x.reshape(B, C, oy, p, ox, p).permute(0, 2, 4, 1, 3, 5).reshape(B, oy * ox, C * p * p)
    w = W["dino.patch_embed.proj.weight"].reshape(-1, C * p * p)
    return linear(cols, w, W["dino.patch_embed.proj.bias"], q)


def prepare_tokens(x: Tensor, W: Dict[str, Tensor], p: int, q: Callable = _ident) -> Tensor:
    B, _, H, Wd = x.shape
    tok = patch_embed(x, W, p, q)
    cls = W["dino.cls_token"].expand(B, -1, -1)
    tok = torch.cat([cls, tok], dim=1)
    if H != Wd:
        raise ValueError("oracle covers square inputs only")
    return tok + resample_pos_embed(W["dino.pos_embed"], H // p)


def attention(x: Tensor, W: Dict[str, Tensor], pre: str, H: int, q: Callable = _ident,
              return_probs: bool = False):
    B, N, D = x.shape
    dh = D // H
    qkv = linear(x, W[pre + "attn.qkv.weight"], W[pre + "attn.qkv.bias"], q)
    qkv = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    Q, K, V = qkv[0], qkv[1], qkv[2]
    s = (q(Q) @ q(K).transpose(-2, -1)) * (dh ** -0.5)      # materialised [B,H,N,N], as the reference
    s = s - s.amax(dim=-1, keepdim=True)
    e = torch.exp(s)
    pr = e / e.sum(dim=-1, keepdim=True)
    ctx = (q(pr) @ q(V)).transpose(1, 2).reshape(B, N, D)
    out = linear(ctx, W[pre + "attn.proj.weight"], W[pre + "attn.proj.bias"], q)
    return (out, pr) if return_probs else out


def mlp(x: Tensor, W: Dict[str, Tensor], pre: str, q: Callable = _ident) -> Tensor:
    h = gelu_erf(linear(x, W[pre + "mlp.fc1.weight"], W[pre + "mlp.fc1.bias"], q))
    return linear(h, W[pre + "mlp.fc2.weight"], W[pre + "mlp.fc2.bias"], q)


def block(x: Tensor, W: Dict[str, Tensor], i: int, H: int, eps: float, q: Callable = _ident) -> Tensor:
    pre = f"dino.blocks.{i}."
    x = x + attention(layer_norm(x, W[pre + "norm1.weight"], W[pre + "norm1.bias"], eps), W, pre, H, q)
    x = x + mlp(layer_norm(x, W[pre + "norm2.weight"], W[pre + "norm2.bias"], eps), W, pre, q)
    return x


def count_blocks(W: Dict[str, Tensor]) -> int:
    n = 0
    while f"dino.blocks.{n}.norm1.weight" in W:
        n += 1
    return n


def vit_forward(x: Tensor, W: Dict[str, Tensor], num_heads: int, patch: int = 8, eps: float = 1e-6,
                q: Callable = _ident, taps: Optional[dict] = None) -> Tensor:
    """[B,3,r,r] fp32 -> [B,N,D] normalised tokens (VisionTransformer.forward, all=True)."""
    t = prepare_tokens(x, W, patch, q)
    if taps is not None:
        taps["tokens"] = t
    for i in range(count_blocks(W)):
        t = block(t, W, i, num_heads, eps, q)
        if taps is not None:
            taps[f"block{i}"] = t
    return layer_norm(t, W["dino.norm.weight"], W["dino.norm.bias"], eps)


def intermediate_layers(x: Tensor, W: Dict[str, Tensor], num_heads: int, n: int = 1, patch: int = 8, eps: float = 1e-6) -> list:
    """norm(x) after each of the last n blocks (VisionTransformer.get_intermediate_layers, vision_transformer.py:282-290)."""
    t = prepare_tokens(x, W, patch)
    L = count_blocks(W)
    out = []
    for i in range(L):
        t = block(t, W, i, num_heads, eps)
        if L - i <= n:
            out.append(layer_norm(t, W["dino.norm.weight"], W["dino.norm.bias"], eps))
    return out


def last_selfattention(x: Tensor, W: Dict[str, Tensor], num_heads: int, patch: int = 8, eps: float = 1e-6) -> Tensor:
    """[B,3,r,r] -> softmax attention [B,H,N,N] of the last block (VisionTransformer.get_last_selfattention,
    vision_transformer.py:273-280)."""
    t = prepare_tokens(x, W, patch)
    L = count_blocks(W)
    for i in range(L - 1):
        t = block(t, W, i, num_heads, eps)
    pre = f"dino.blocks.{L - 1}."
    _, pr = attention(layer_norm(t, W[pre + "norm1.weight"], W[pre + "norm1.bias"], eps), W, pre, num_heads, return_probs=True)
    return pr


def _masked_cls_attention(t: Tensor, W: Dict[str, Tensor], pre: str, H: int, cls_mask: Tensor):
    """Attention.forward with cls_mask (vision_transformer.py:80-107): only the CLS query row is kept, its logits are
    MULTIPLIED by each of the Nm masks (a zero column is prepended for the CLS key, so masked keys keep logit 0 rather
    than -inf), softmax, then attn @ v gives one context row per mask.  t: [1, N, D] (already norm1'd); cls_mask
    [Nm, ph, pw].  Returns (proj output [1, Nm, D], probabilities [1, H, Nm, N])."""
    B, N, D = t.shape
    dh = D // H
    qkv = linear(t, W[pre + "attn.qkv.weight"], W[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    Q, K, V = qkv[0], qkv[1], qkv[2]
    s = (Q @ K.transpose(-2, -1)) * (dh ** -0.5)
    Nm = cls_mask.shape[0]
    m = torch.cat([torch.zeros((Nm, 1), dtype=s.dtype), cls_mask.reshape(Nm, -1).to(s.dtype)], dim=1)    # [Nm, N]
    s = s[0:1, :, 0:1, :] * m                                                                              # [1, H, Nm, N]
    s = s - s.amax(dim=-1, keepdim=True)
    e = torch.exp(s)
    pr = e / e.sum(dim=-1, keepdim=True)
    ctx = (pr @ V).transpose(1, 2).reshape(B, Nm, D)
    return linear(ctx, W[pre + "attn.proj.weight"], W[pre + "attn.proj.bias"]), pr


def forward_mask(x: Tensor, W: Dict[str, Tensor], num_heads: int, cls_mask: Tensor, patch: int = 8, eps: float = 1e-6,
                 return_attention: bool = False) -> Tensor:
    """VisionTransformer.forward_mask (vision_transformer.py:250-271) / get_last_selfattention(x, cls_mask) (:273-280) for
    one frame x [1,3,r,r]: all blocks but the last run normally; in the last one the CLS token attends through each of the
    Nm masks and the CLS residual is repeated Nm times (Block.forward :127-140).  Returns the Nm mask embeddings [Nm, D]
    after the final norm, or the masked attention [1, H, Nm, N]."""
    t = prepare_tokens(x, W, patch)
    L = count_blocks(W)
    for i in range(L - 1):
        t = block(t, W, i, num_heads, eps)
    pre = f"dino.blocks.{L - 1}."
    y, pr = _masked_cls_attention(layer_norm(t, W[pre + "norm1.weight"], W[pre + "norm1.bias"], eps), W, pre, num_heads, cls_mask)
    if return_attention:
        return pr
    xm = t[:, 0:1, :].repeat(1, cls_mask.shape[0], 1) + y
    xm = xm + mlp(layer_norm(xm, W[pre + "norm2.weight"], W[pre + "norm2.bias"], eps), W, pre)
    return layer_norm(xm, W["dino.norm.weight"], W["dino.norm.bias"], eps)[0]


def head_forward(feat: Tensor, W: Dict[str, Tensor], q: Callable = _ident) -> Tensor:
    """[M,D] -> [M,C] log-probabilities; MLP head if clf.layer_2 exists, else the Linear head."""
    if "clf.layer_2.weight" in W:
        h = torch.relu(linear(feat, W["clf.layer_1.weight"], W["clf.layer_1.bias"], q))
        h = torch.relu(linear(h, W["clf.layer_2.weight"], W["clf.layer_2.bias"], q))
        z = linear(h, W["clf.layer_3.weight"], W["clf.layer_3.bias"], q)
    else:
        z = linear(feat, W["clf.layer_1.weight"], W["clf.layer_1.bias"], q)
    z = z - z.amax(dim=1, keepdim=True)
    return z - torch.log(torch.exp(z).sum(dim=1, keepdim=True))


def dinoseg_forward(x: Tensor, W: Dict[str, Tensor], num_heads: int, patch: int = 8, eps: float = 1e-6,
                    q: Callable = _ident) -> Tensor:
    """[B,3,r,r] -> [B*n, C] log-probs: ViT, drop CLS, flatten patches, head."""
    t = vit_forward(x, W, num_heads, patch, eps, q)[:, 1:]
    return head_forward(t.reshape(-1, t.shape[-1]), W)


def predict(frame_u8: np.ndarray, W: Dict[str, Tensor], num_heads: int, resolution: int,
            patch: int = 8) -> np.ndarray:
    """uint8 [r,r,3] frame (already at `resolution`) -> int64 map, reference predict() semantics.

    argmax takes the first maximum; the low-res o x o map is replicated by np.kron with a
    (480//o) x (480//o) block of ones -- including the r=400 -> 450x450 quirk (pl_torch_modules.py:294-298).
    """
    if resolution % 8 != 0:
        raise ValueError("Resolution should be a multiple of 8.")
    with torch.no_grad():
        lp = dinoseg_forward(preprocess(frame_u8[None]), W, num_heads, patch)
    o = resolution // 8
    low = torch.argmax(lp, dim=-1).numpy().reshape(o, o)
    k = 480 // o
    return np.kron(low, np.ones((k, k), dtype=int))


def nll_loss(logp: Tensor, labels: Tensor, ignore_index: int = -100) -> Tensor:
    """F.nll_loss with its defaults (pl_torch_modules.py:265): mean negative log-likelihood over the patches whose label is
    not `ignore_index` (-100); any other label outside [0, C) is an error, as in torch."""
    y = labels.reshape(-1).long()
    keep = y != ignore_index
    if bool(((y < 0) | (y >= logp.shape[1]))[keep].any()):
        raise IndexError("Target out of bounds")
    picked = -logp.gather(1, y.clamp(min=0).reshape(-1, 1)).reshape(-1)
    return (picked * keep).sum() / keep.sum()


def to_torch(state: Dict[str, np.ndarray], requires_grad: bool = False) -> Dict[str, Tensor]:
    out = {}
    for k, v in state.items():
        t = torch.from_numpy(np.array(v, dtype=np.float32, copy=True))
        out[k] = t.requires_grad_(True) if requires_grad else t
    return out


# --------------------------------------------------------------------------- precision emulation
def quant_bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def quant_fp16(x: Tensor) -> Tensor:
    return x.to(torch.float16).to(torch.float32)


def quant_bf16x2(x: Tensor) -> Tensor:
    """hi + lo split: what a 3-MFMA bf16 split product effectively sees of each operand."""
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi + lo
