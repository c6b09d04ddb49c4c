"""Host-side mirror of the reference ``DINOSeg`` (dt_segmentation/src/pl_torch_modules.py:141-440).

Same public surface -- ``load_from_checkpoint()``, ``.to()``, ``set_resolution()``, ``predict()``,
``forward()``/``__call__``, ``state_dict()`` keys, ``freeze_bb()/unfreeze_bb()`` -- with every
tensor op replaced by the hand-written gfx950 kernels behind the C-ABI in ``include/dinoseg.h``.
PyTorch only owns the parameters, the I/O tensors and the stream.  There is no CPU path: calling
the model without a ROCm device raises.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import capi
from .preprocess import resize_linear_u8
from .weights import VIT_B8, VIT_S8, ViTConfig

_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)
_PRECISIONS = {"bf16": capi.BF16, "bf16x3": capi.BF16X3, "fp16": capi.FP16, "fp16x3": capi.FP16X3}
# precision "auto" (the default): the native handle of a call is chosen by what the call needs -- inference (predict,
# forward_frames, forward under no_grad or with nothing trainable, the visualisation paths) runs fp16 hi+lo planes, the parity mode
# that holds the flat 1e-3 bar with margin; a call that produces gradients runs bf16 hi+lo planes, the parity mode that trains
_AUTO_INFER, _AUTO_TRAIN = "fp16x3", "bf16x3"
# per-handle state of the model object: one set per native handle (precision "auto" keeps up to two)
_SLOT_KEYS = ("_handle", "_bound_sig", "_grad_sig", "_ptr_key", "_fast_index", "_fast_sig", "_pred_graphs")


# --------------------------------------------------------------------------- preprocessing mirror
class _Transforms:
    """Call-compatible stand-in for the albumentations pipeline of ``get_transforms``
    (pl_torch_modules.py:33-41): ``t(image=ndarray)['image']`` -> fp32 CHW tensor.

    Resize(res, res) is the identity for frames already at res x res (the benchmark / golden case);
    other sizes go through cv2.INTER_LINEAR's fixed-point arithmetic restated in dino_amd/preprocess.py
    (parity unpinned: albumentations / OpenCV are not available here -- see DESIGN.md).  ``predict`` does the same
    arithmetic on the GPU (``dinoseg_op_resize_u8``) and never calls this host path.
    """

    def __init__(self, resolution: int):
        self.resolution = int(resolution)

    def resize(self, img: np.ndarray) -> np.ndarray:
        r = self.resolution
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError(f"expected an HxWx3 image, got {img.shape}")
        if img.shape[0] == r and img.shape[1] == r:
            return np.ascontiguousarray(img, dtype=np.uint8)
        return resize_linear_u8(img, r, r)

    def __call__(self, image: np.ndarray) -> Dict[str, torch.Tensor]:
        u8 = self.resize(np.asarray(image))
        mean = np.array(_IMAGENET_MEAN, dtype=np.float32) * np.float32(255.0)
        inv = np.reciprocal(np.array(_IMAGENET_STD, dtype=np.float32) * np.float32(255.0))
        x = (u8.astype(np.float32) - mean) * inv
        return {"image": torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))}


def get_transforms(resolution: int = 480) -> _Transforms:
    return _Transforms(resolution)


def metrics_from_confusion(cm: np.ndarray, prefix: str = "val") -> Dict[str, float]:
    """cm[gt][pred] counts -> {prefix_acc (balanced accuracy), prefix_iou (macro Jaccard), prefix_F1 (macro F1)}."""
    tp = np.diag(cm)
    support = cm.sum(axis=1)          # per true class
    predicted = cm.sum(axis=0)        # per predicted class
    present = support > 0
    acc = float(np.mean(tp[present] / support[present])) if present.any() else 0.0
    labels = (support + predicted) > 0                       # sklearn: union of the labels seen in y_true and y_pred
    denom_f1 = 2 * tp + (predicted - tp) + (support - tp)
    denom_iou = tp + (predicted - tp) + (support - tp)
    f1 = float(np.mean(np.where(denom_f1[labels] > 0, 2 * tp[labels] / np.maximum(denom_f1[labels], 1), 0.0))) if labels.any() else 0.0
    iou = float(np.mean(np.where(denom_iou[labels] > 0, tp[labels] / np.maximum(denom_iou[labels], 1), 0.0))) if labels.any() else 0.0
    return {prefix + "_acc": acc, prefix + "_iou": iou, prefix + "_F1": f1}


# --------------------------------------------------------------------------- parameter containers
class _Attn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.qkv = nn.Linear(D, 3 * D, bias=True)
        self.proj = nn.Linear(D, D)


class _Mlp(nn.Module):
    def __init__(self, D, F):
        super().__init__()
        self.fc1 = nn.Linear(D, F)
        self.fc2 = nn.Linear(F, D)


class _Block(nn.Module):
    def __init__(self, D, F, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=eps)
        self.attn = _Attn(D)
        self.norm2 = nn.LayerNorm(D, eps=eps)
        self.mlp = _Mlp(D, F)


class _PatchEmbed(nn.Module):
    def __init__(self, D, p):
        super().__init__()
        self.proj = nn.Conv2d(3, D, kernel_size=p, stride=p)


class _ViTParams(nn.Module):
    """The backbone as the reference exposes it (``model.dino``): the parameter names of the reference ViT's state_dict
    (vision_transformer.py:161-196) and its call surface -- ``dino(x)``, ``dino.get_last_selfattention(x)``,
    ``dino.forward_mask(x, cls_mask)`` -- with the arithmetic in libdinoseg_hip.so.  It holds a weak reference to the
    owning DINOSeg (which owns the native handle); a detached copy raises instead of computing on stale state."""

    def __init__(self, cfg: ViTConfig):
        super().__init__()
        D = cfg.embed_dim
        self.patch_embed = _PatchEmbed(D, cfg.patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, cfg.pos_grid * cfg.pos_grid + 1, D))
        self.blocks = nn.ModuleList([_Block(D, cfg.hidden, cfg.ln_eps) for _ in range(cfg.n_blocks)])
        self.norm = nn.LayerNorm(D, eps=cfg.ln_eps)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        object.__setattr__(self, "_owner_ref", None)

    def _init_weights(self, m):
        """vision_transformer.py:189-196 (what ``random_init=True`` re-applies, pl_torch_modules.py:181-183)."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _set_owner(self, owner) -> None:
        object.__setattr__(self, "_owner_ref", weakref.ref(owner))

    def _owner(self):
        ref = self.__dict__.get("_owner_ref")
        owner = ref() if ref is not None else None
        if owner is None:
            raise capi.DinosegError("this backbone is not attached to a DINOSeg (the native handle lives there)")
        return owner

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_owner_ref"] = None          # weak references do not pickle / deep-copy; DINOSeg re-links its copy
        return state

    def forward(self, x: torch.Tensor, all: bool = True, intermediate=False) -> torch.Tensor:
        """VisionTransformer.forward (vision_transformer.py:237-248): fp32 [B,3,r,r] -> final-norm tokens [B, N, D]
        (``all=False``: the CLS row [B, D]); ``intermediate=k`` returns ``norm(x)`` of ALL tokens after block k, whatever ``all``
        says (:241-242) -- unless k is past the last block, where the loop never exits early and ``all`` applies (:243-248)."""
        owner = self._owner()
        k = int(intermediate) if intermediate else 0
        early = 0 < k <= owner.n_blocks
        t = owner.features(x, n_blocks=k if early else 0)
        return t if (all or early) else t[:, 0]

    def get_last_selfattention(self, x, cls_mask=None):
        """vision_transformer.py:273-280; reference call site visualize_attention.py:46."""
        return self._owner().get_last_selfattention(x, cls_mask)

    def get_intermediate_layers(self, x: torch.Tensor, n: int = 1) -> list:
        """vision_transformer.py:282-290: ``norm(x)`` of all tokens after each of the last ``n`` blocks, oldest first (a list of
        [B, N, D] tensors; every block when n exceeds the depth).  One ``dinoseg_features`` call per tap: the library keeps one
        residual stream, so the forward is re-run up to the tapped block (n is 1 in every published use of this method)."""
        owner = self._owner()
        L = owner.cfg.n_blocks
        return [owner.features(x, n_blocks=i + 1) for i in range(L) if L - i <= n]

    def forward_mask(self, x, cls_mask):
        """vision_transformer.py:250-271."""
        return self._owner().forward_mask(x, cls_mask)


class _MLPHead(nn.Module):
    def __init__(self, n_classes, input_dim):
        super().__init__()
        self.layer_1 = nn.Linear(input_dim, 200)
        self.layer_2 = nn.Linear(200, 100)
        self.layer_3 = nn.Linear(100, n_classes)


class _LinearHead(nn.Module):
    def __init__(self, n_classes, input_dim):
        super().__init__()
        self.layer_1 = nn.Linear(input_dim, n_classes)


class _DinoSegFunction(torch.autograd.Function):
    """DINOSeg.forward under autograd: forward = dinoseg_train_forward (activations kept in the library's workspace),
    backward = dinoseg_backward(d loss / d logp).  The parameters are inputs only so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, model, x, kind, B, r, *params):
        logp = model._autograd_forward(x, kind, B, r)
        ctx.model = model
        ctx.epoch = model._fwd_epoch
        ctx.params = params
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        grads = ctx.model._autograd_backward(dlogp, ctx.epoch, ctx.params)
        return (None, None, None, None, None) + grads


# --------------------------------------------------------------------------- the model
class DINOSeg(nn.Module):
    """DINO ViT + per-patch segmentation head on MI355X.

    Constructor keywords follow the reference (pl_torch_modules.py:144-147); two extra
    keyword-only arguments select what the reference hard-codes or cannot express:
    ``arch`` ('vit_small' | 'vit_base' | a ViTConfig giving embed_dim/num_heads) and ``precision``: 'auto' (default: 'fp16x3'
    for inference calls, 'bf16x3' when a gradient is requested), 'fp16x3' / 'bf16x3' (parity modes: hi + lo operand planes),
    'fp16' / 'bf16' (benchmark modes: one plane).
    """

    def __init__(self, data_path=None, write_path=None, class_names=None, head="linear", n_blocks=1,
                 batch_size=1, lr=1e-6, optimizer=torch.optim.AdamW, freeze_backbone=True, max_epochs=200,
                 patience=10, grayscale=False, n_classes=7, pretrain_on_sim=False, comet_logger=None,
                 augmented=True, random_init=False, backbone="vit", *, arch="vit_small", precision="auto"):
        super().__init__()
        if backbone != "vit":
            raise NotImplementedError("only backbone='vit' is on the MI355X hot path (SURVEY.md §2 row 7)")
        if head not in ("linear", "mlp"):
            raise ValueError(f"unknown head {head!r}")
        if precision != "auto" and precision not in _PRECISIONS:
            raise ValueError(f"precision must be 'auto' or one of {sorted(_PRECISIONS)}")
        base = arch if isinstance(arch, ViTConfig) else {"vit_small": VIT_S8, "vit_base": VIT_B8}[arch]
        self.cfg = ViTConfig(embed_dim=base.embed_dim, num_heads=base.num_heads, n_blocks=int(n_blocks),
                             n_classes=int(n_classes), head=head)
        self.arch = arch
        self.precision = precision
        self.n_blocks = n_blocks
        self.head = head
        self.batch_size = batch_size
        self.lr = lr
        self.optimizer = optimizer
        self.freeze_backbone = freeze_backbone
        self.max_epochs = max_epochs
        self.patience = patience
        self.grayscale = grayscale
        self.n_classes = n_classes
        self.comet_logger = comet_logger
        self.class_names = class_names
        self.pretrain_on_sim = pretrain_on_sim
        self.augmented = augmented
        self.random_init = random_init
        self.backbone = backbone
        self.mlp_input_dim = self.cfg.embed_dim
        self.data_path, self.write_path = data_path, write_path
        self.best_ck = None

        self.resolution = 480
        self.transforms = get_transforms(self.resolution)

        # No network: the pretrained DINO fetch of the reference (dt_utils.py:19-29) is replaced by
        # a random init; real weights arrive through load_state_dict / load_from_checkpoint.
        self.dino = _ViTParams(self.cfg)
        self.clf = (_MLPHead(self.cfg.n_classes, self.cfg.embed_dim) if head == "mlp"
                    else _LinearHead(self.cfg.n_classes, self.cfg.embed_dim))

        self.dino._set_owner(self)
        if random_init:         # pl_torch_modules.py:181-183
            self.dino.apply(self.dino._init_weights)

        self._handle: Optional[C.c_void_p] = None
        self._bound_sig = None
        self._grad_sig = None
        self._weights_epoch = 0

    # ---- plumbing -------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self.dino.cls_token.device

    def _require_gpu(self) -> None:
        if self.device.type != "cuda":
            raise capi.DinosegError("DINOSeg runs only on a ROCm device (call .to('cuda:0')); there is no CPU path")

    def _param_signature(self):
        return (self._weights_epoch,) + tuple((k, v.data_ptr(), v._version) for k, v in self.state_dict(keep_vars=True).items())

    # The full signature walks state_dict() (~0.2 ms for 156 tensors): a 20-40 % tax on a single-frame predict().  The fast check
    # below sees the same events without building it: every tensor's version counter and address (in-place updates, optimizer
    # steps, load_state_dict, .to()), the identity of every entry of every module's parameter / buffer / child table (a replaced
    # Parameter or sub-module), and the tables' sizes (an added one).  Only when it differs is the full signature rebuilt.
    def _build_fast_index(self):
        tens, tables = [], []
        for mod in self.modules():
            for d in (mod._parameters, mod._buffers, mod._modules):
                tables.append((d, len(d)))
                tens.extend((d, k, t) for k, t in d.items() if t is not None)
        return tens, tables

    def _fast_signature(self):
        idx = self.__dict__.get("_fast_index")
        if idx is None:
            return None
        tens, tables = idx
        for d, n in tables:
            if len(d) != n:
                return None
        vers = []
        for d, k, t in tens:
            if d.get(k) is not t:
                return None
            if isinstance(t, torch.Tensor):
                vers.append(t._version)
                vers.append(t.data_ptr())
        vers.append(self._weights_epoch)
        return vers

    def invalidate_weights(self) -> None:
        """Force a re-pack of the bf16 weight planes on the next call.  In-place edits through ``p.data`` (or any write that
        does not bump the tensor version counter) are invisible to the automatic check; call this after them."""
        self._weights_epoch += 1

    def _stream(self) -> int:
        """hipStream_t of torch's current stream ON THE MODEL'S DEVICE (not the caller's current device)."""
        return capi.stream_ptr(self.device)

    # the native handle, its bound-pointer signatures and the weak owner link are process state, not model state
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"] = None
        state["_bound_sig"] = None
        state["_grad_sig"] = None
        state.pop("_fast_index", None)
        state.pop("_fast_sig", None)
        state.pop("_pred_graphs", None)
        state.pop("_ptr_key", None)
        state.pop("_adam_state", None)
        state.pop("_grad_bucket_cache", None)
        state.pop("_slots", None)
        state.pop("_active_precision", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._handle = None
        self._bound_sig = None
        self._grad_sig = None
        self.dino._set_owner(self)

    def _use(self, train: bool) -> None:
        """precision 'auto': make the native handle of the wanted kind the active one (created on first use by _sync_weights).
        Each handle keeps its own bound-pointer signatures, packed weights, workspaces and captured predict() graphs, so an
        evaluation between two training steps costs no re-creation; both read the SAME parameter tensors and re-pack on their
        own next use after an update."""
        if self.precision != "auto":
            return
        want = _AUTO_TRAIN if train else _AUTO_INFER
        cur = self.__dict__.get("_active_precision")
        if cur == want:
            return
        slots = self.__dict__.setdefault("_slots", {})
        if cur is not None:
            slots[cur] = {k: self.__dict__.pop(k, None) for k in _SLOT_KEYS}
        st = slots.pop(want, None) or {}
        for k in _SLOT_KEYS:
            v = st.get(k)
            if v is None and k not in ("_handle", "_bound_sig", "_grad_sig"):
                self.__dict__.pop(k, None)
            else:
                self.__dict__[k] = v
        self._active_precision = want

    def effective_precision(self, train: bool = False) -> str:
        """The operand precision a call of the given kind runs in ('auto' resolved)."""
        if self.precision != "auto":
            return self.precision
        return _AUTO_TRAIN if train else _AUTO_INFER

    def _sync_weights(self, train: bool = False) -> None:
        """Create the native handle if needed and (re)bind + repack when any parameter moved or changed."""
        lib = capi.lib()
        self._require_gpu()
        self._use(train)
        if self._handle is None:
            cfg = capi.Config(self.cfg.embed_dim, self.cfg.num_heads, self.cfg.n_blocks, self.cfg.patch,
                              self.cfg.mlp_ratio, self.cfg.n_classes,
                              capi.HEAD_MLP if self.head == "mlp" else capi.HEAD_LINEAR, self.cfg.pos_grid,
                              self.cfg.ln_eps, _PRECISIONS[self.effective_precision(train)])
            h = C.c_void_p()
            capi.check(lib.dinoseg_create(C.byref(cfg), C.byref(h)))
            self._handle = h
            self._bound_sig = None
        if self._bound_sig is not None:
            fast = self._fast_signature()
            if fast is not None and fast == self.__dict__.get("_fast_sig"):
                return
        sig = self._param_signature()
        if sig == self._bound_sig:
            self._fast_index = self._build_fast_index()
            self._fast_sig = self._fast_signature()
            return
        for name, t in self.state_dict(keep_vars=True).items():
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise capi.DinosegError(f"parameter {name} must be contiguous fp32")
            shape = (C.c_int64 * t.dim())(*t.shape)
            capi.check(lib.dinoseg_bind_weight(self._handle, name.encode(), t.data_ptr(), shape, t.dim()))
        capi.check(lib.dinoseg_refresh_weights(self._handle, self._stream()))
        self._bound_sig = sig
        self._ptr_key = (id(self._handle),) + tuple(e[1] for e in sig[1:])      # parameter storage the library's kernels read directly
        self._fast_index = self._build_fast_index()
        self._fast_sig = self._fast_signature()

    def set_precision(self, precision: str) -> None:
        if precision != "auto" and precision not in _PRECISIONS:
            raise ValueError(f"precision must be 'auto' or one of {sorted(_PRECISIONS)}")
        if precision != self.precision:
            self._release()
            self.precision = precision

    def _release(self) -> None:
        self.__dict__.pop("_pred_graphs", None)          # captured graphs hold the handle's workspace and packed-weight addresses
        for st in self.__dict__.pop("_slots", {}).values():      # precision 'auto': the handle that is not the active one
            if st.get("_handle") is not None:
                capi.lib().dinoseg_destroy(st["_handle"])
        self.__dict__.pop("_active_precision", None)
        if self._handle is not None:
            capi.lib().dinoseg_destroy(self._handle)
            self._handle = None
            self._bound_sig = None
            self._grad_sig = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ---- reference API ----------------------------------------------------------------------
    def set_resolution(self, resolution=480):
        if resolution % 8 != 0:
            raise ValueError("Resolution should be a multiple of 8.")
        self.transforms = get_transforms(resolution)
        self.resolution = resolution

    def _run(self, x: torch.Tensor, kind: int, B: int, r: int, want_logp: bool = True, want_argmax: bool = False,
             tap_block: int = -1):
        self._sync_weights()
        n = (r // 8) ** 2
        dev = x.device
        logp = torch.empty((B * n, self.cfg.n_classes), dtype=torch.float32, device=dev) if want_logp else None
        amax = torch.empty((B * n,), dtype=torch.int32, device=dev) if want_argmax else None
        tap = (torch.empty((B * (n + 1), self.cfg.embed_dim), dtype=torch.float32, device=dev)
               if tap_block >= 0 else None)
        capi.check(capi.lib().dinoseg_forward(self._handle, x.data_ptr(), kind, B, r, capi.ptr(logp), capi.ptr(amax),
                                              tap_block, capi.ptr(tap), self._stream()))
        return logp, amax, tap

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 [B,3,r,r] (normalised) -> fp32 [B*(r/8)^2, n_classes] log-probabilities (pl_torch_modules.py:239-256).
        With grad enabled and at least one trainable parameter the result carries an autograd graph: its backward calls
        ``dinoseg_backward`` and hands d loss / d parameter to torch (x itself gets no gradient: the reference never asks)."""
        self._require_gpu()
        x, kind, B, r = self._prep_batch(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _DinoSegFunction.apply(self, x, kind, B, r, *self.parameters())
        logp, _, _ = self._run(x, kind, B, r)
        return logp

    def features(self, x: torch.Tensor, n_blocks: int = 0) -> torch.Tensor:
        """``model.dino(x)``: final-norm tokens fp32 [B, N, D] after `n_blocks` blocks (0 = all), CLS token first
        (VisionTransformer.forward, vision_transformer.py:237-248).  Inference only (no autograd graph)."""
        self._require_gpu()
        x, kind, B, r = self._prep_batch(x)
        if not 0 <= n_blocks <= self.cfg.n_blocks:
            raise ValueError(f"intermediate must be in [0, {self.cfg.n_blocks}]")
        self._sync_weights()
        out = torch.empty((B, (r // 8) ** 2 + 1, self.cfg.embed_dim), dtype=torch.float32, device=self.device)
        capi.check(capi.lib().dinoseg_features(self._handle, x.data_ptr(), kind, B, r, n_blocks, out.data_ptr(), self._stream()))
        return out

    @torch.no_grad()
    def forward_frames(self, frames_u8: torch.Tensor, want_logp: bool = True):
        """uint8 [B,r,r,3] device frames -> (log-probs or None, int32 argmax [B*(r/8)^2]).
        The batched form of predict(): normalisation is fused into the patch gather on device."""
        self._require_gpu()
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3 \
                or frames_u8.shape[1] != frames_u8.shape[2]:
            raise ValueError(f"expected uint8 [B,r,r,3], got {frames_u8.dtype} {tuple(frames_u8.shape)}")
        if frames_u8.shape[1] % 8 != 0:
            raise ValueError("Resolution should be a multiple of 8.")
        frames_u8 = frames_u8.to(self.device).contiguous()
        logp, amax, _ = self._run(frames_u8, capi.INPUT_U8_HWC, frames_u8.shape[0], frames_u8.shape[1],
                                  want_logp=want_logp, want_argmax=True)
        return logp, amax

    def _predict_graph(self, r: int):
        """The single-frame forward of ``predict()`` as a captured HIP graph (one replay instead of ~150 launches: a 12-block
        forward is 1.33 ms of kernels that the eager launch path stretches to 1.50).  Static input / output buffers; captured once
        per (resolution, bound parameter storage) -- an in-place weight update re-packs into the same library buffers before the
        replay, new parameter storage (``.to()``, ``load_state_dict`` of new tensors) re-captures.  Returns None where capture is
        not possible (``predict_graph = False``, a capture already in progress, a failed capture: eager from then on)."""
        if not getattr(self, "predict_graph", True) or torch.cuda.is_current_stream_capturing():
            return None
        cache = self.__dict__.setdefault("_pred_graphs", {})
        ent = cache.get(r)
        if ent is not None and ent["key"] == self._ptr_key and capi.lib().dinoseg_state_generation(self._handle) == ent["gen"]:
            return ent
        try:
            dev = self.device
            static_in = torch.zeros((1, r, r, 3), dtype=torch.uint8, device=dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self.forward_frames(static_in, want_logp=False)          # warm-up: workspace, resolution cache, packs
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    _, static_out = self.forward_frames(static_in, want_logp=False)
            torch.cuda.current_stream(dev).wait_stream(side)
            ent = {"key": self._ptr_key, "graph": g, "in": static_in, "out": static_out,
                   "gen": capi.lib().dinoseg_state_generation(self._handle)}
            cache.clear()           # (one resolution at a time: the library caches one resampled position embedding)
            cache[r] = ent
            return ent
        except Exception:
            self.predict_graph = False          # (e.g. a runtime without graph support for one of the calls): eager from now on
            return None

    def predict(self, x) -> np.ndarray:
        """Run inference on a single image (PIL.Image or HxWx3 uint8 array); returns the int64 map the
        reference returns: np.kron of the (r/8)x(r/8) argmax map with a (480//(r/8))^2 block of ones
        (pl_torch_modules.py:276-300, including the non-480 sizes it yields when 480 % (r/8) != 0)."""
        with torch.no_grad():
            raw = np.asarray(x)
            if raw.dtype != np.uint8 or not raw.flags.c_contiguous:
                raw = np.ascontiguousarray(raw, dtype=np.uint8)
            if raw.ndim != 3 or raw.shape[2] != 3:
                raise ValueError(f"expected an HxWx3 image, got {raw.shape}")
            r = self.resolution
            self._require_gpu()
            self._sync_weights()
            ent = self._predict_graph(r)
            frames = torch.from_numpy(raw).unsqueeze(0)                           # uint8 on the wire, whatever its size
            if raw.shape[0] != r or raw.shape[1] != r:                           # Resize(r, r) of get_transforms, on the GPU
                frames = frames.to(self.device)
                resized = ent["in"] if ent is not None else torch.empty((1, r, r, 3), dtype=torch.uint8, device=self.device)
                capi.check(capi.lib().dinoseg_op_resize_u8(frames.data_ptr(), raw.shape[0], raw.shape[1], resized.data_ptr(), r, r,
                                                          self._stream()))
                frames = resized
            elif ent is not None:
                ent["in"].copy_(frames, non_blocking=True)
            if ent is not None:
                ent["graph"].replay()
                amax = ent["out"]
            else:
                _, amax = self.forward_frames(frames.to(self.device), want_logp=False)
            output_size = self.resolution // 8
            low_res = amax.cpu().numpy().astype(np.int64).reshape((output_size, output_size))
            high_res_patch_size = 480 // output_size
            # == np.kron(low_res, np.ones((k, k), dtype=int)) of the reference (:297-298), 4x cheaper on the host
            k = high_res_patch_size
            return np.repeat(np.repeat(low_res, k, axis=0), k, axis=1)

    def debug_tokens(self, x: torch.Tensor, block: int) -> torch.Tensor:
        """Token matrix [B, N, D] after prepare_tokens (block=0) or after transformer block `block`."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        _, _, tap = self._run(x, capi.INPUT_F32_CHW, x.shape[0], x.shape[2], tap_block=block)
        return tap.reshape(x.shape[0], -1, self.cfg.embed_dim)

    def _mask_request(self, x: torch.Tensor, cls_mask: torch.Tensor, want_emb: bool, want_attn: bool):
        self._require_gpu()
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape[0] != 1 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] % 8 != 0:
            raise ValueError(f"expected a single frame [1,3,r,r] with r % 8 == 0, got {tuple(x.shape)}")
        r = x.shape[2]
        n = (r // 8) ** 2
        m = cls_mask.to(device=self.device, dtype=torch.float32).reshape(cls_mask.shape[0], -1).contiguous()
        if m.shape[1] != n:
            raise ValueError(f"cls_mask must be [n_masks, {r // 8}, {r // 8}], got {tuple(cls_mask.shape)}")
        self._sync_weights()
        emb = torch.empty((m.shape[0], self.cfg.embed_dim), dtype=torch.float32, device=x.device) if want_emb else None
        att = torch.empty((1, self.cfg.num_heads, m.shape[0], n + 1), dtype=torch.float32, device=x.device) if want_attn else None
        capi.check(capi.lib().dinoseg_forward_mask(self._handle, x.data_ptr(), capi.INPUT_F32_CHW, r, m.data_ptr(), m.shape[0],
                                                   capi.ptr(emb), capi.ptr(att), self._stream()))
        return emb, att

    def forward_mask(self, x: torch.Tensor, cls_mask: torch.Tensor) -> torch.Tensor:
        """One embedding per mask, [n_masks, embed_dim]: ``model.dino.forward_mask(x, cls_mask)`` of the reference
        (vision_transformer.py:250-271).  x: one frame fp32 [1,3,r,r]; cls_mask [n_masks, r/8, r/8]."""
        return self._mask_request(x, cls_mask, True, False)[0]

    def get_last_selfattention(self, x: torch.Tensor, cls_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Attention probabilities [B, heads, N, N] of the last block (reference: ``model.dino.get_last_selfattention(x)``,
        vision_transformer.py:273-280; used by visualize_attention.py:46).  x: fp32 [B,3,r,r].  With cls_mask
        [n_masks, r/8, r/8]: the masked CLS attention [1, heads, n_masks, N] of one frame."""
        if cls_mask is not None:
            return self._mask_request(x, cls_mask, False, True)[1]
        self._require_gpu()
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3] or x.shape[2] % 8 != 0:
            raise ValueError(f"expected [B,3,r,r] with r % 8 == 0, got {tuple(x.shape)}")
        self._sync_weights()
        B, r = x.shape[0], x.shape[2]
        N = (r // 8) ** 2 + 1
        out = torch.empty((B, self.cfg.num_heads, N, N), dtype=torch.float32, device=x.device)
        capi.check(capi.lib().dinoseg_last_selfattention(self._handle, x.data_ptr(), capi.INPUT_F32_CHW, B, r, out.data_ptr(),
                                                         self._stream()))
        return out

    # ---- validation metrics (pl_torch_modules.py:302-345) ------------------------------------------
    def validation_step(self, batch, batch_idx=0):
        """Reference: probs = self(x); pred = argmax.  Here the per-batch confusion matrix is accumulated on device."""
        x, y = batch
        with torch.no_grad():
            xx, kind, B, r = self._prep_batch(x)
            logp, amax, _ = self._run(xx, kind, B, r, want_logp=True, want_argmax=True)
            y = y.to(self.device).reshape(-1).long().contiguous()
            cm = torch.zeros((self.cfg.n_classes, self.cfg.n_classes), dtype=torch.int64, device=self.device)
            capi.check(capi.lib().dinoseg_op_confusion(amax.data_ptr(), y.data_ptr(), y.numel(), self.cfg.n_classes,
                                                       cm.data_ptr(), self._stream()))
        return {"pred": amax, "gt": y, "probs": logp, "confusion": cm}

    def validation_epoch_end(self, outputs, prefix="val"):
        """Balanced accuracy, macro F1 and macro IoU over all patches of the split, from the summed confusion matrices
        (same definitions as sklearn's balanced_accuracy_score / f1_score(macro) / jaccard_score(macro), which the
        reference calls on the concatenated predictions, pl_torch_modules.py:317-319)."""
        cm = torch.stack([o["confusion"] for o in outputs]).sum(0).cpu().numpy().astype(np.float64)
        return metrics_from_confusion(cm, prefix)

    def test_step(self, batch, batch_idx=0):
        return self.validation_step(batch, batch_idx)

    def test_epoch_end(self, outputs):
        return self.validation_epoch_end(outputs, prefix="test")

    def profile(self, level: int) -> None:
        """Per-kernel-class HIP-event timing inside forward (0 off, 1 attention only, 2 all classes)."""
        self._sync_weights()
        capi.check(capi.lib().dinoseg_profile(self._handle, int(level)))

    def profile_read(self) -> Dict[str, tuple]:
        """{class name: (summed ms, launches)} since the last read; waits for the recorded events."""
        n = len(capi.PROF_CLASSES)
        ms, cnt = (C.c_float * n)(), (C.c_int32 * n)()
        capi.check(capi.lib().dinoseg_profile_read(self._handle, ms, cnt))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(capi.PROF_CLASSES)}

    # ---- fine-tune step --------------------------------------------------------------------------
    @staticmethod
    def grad_stage(name: str, n_blocks: int) -> int:
        """Backward stage after which the gradient of parameter `name` is final: 0 = head, 1 + k = final norm and block
        n_blocks-1-k, n_blocks + 1 = embeddings (the order dinoseg_backward produces them; see dinoseg_stream_wait_grad_stage)."""
        if name.startswith("clf."):
            return 0
        if name.startswith("dino.norm."):
            return 1 if n_blocks > 0 else 1
        if name.startswith("dino.blocks."):
            return 1 + (n_blocks - 1 - int(name.split(".")[2]))
        return n_blocks + 1

    def _grad_buckets(self, slot: str, bucket_bytes: int = 8 << 20):
        """Flat fp32 gradient buckets (dino_amd.parallel.make_flat_buckets) of the trainable parameters, cached per slot until
        the trainable set, the device or the bucket size changes."""
        from .parallel import make_flat_buckets
        params = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        key = (bucket_bytes,) + tuple((n, p.numel(), str(p.device)) for n, p in params)
        cache = self.__dict__.setdefault("_grad_bucket_cache", {})
        if slot not in cache or cache[slot]["key"] != key:
            cache[slot] = make_flat_buckets(params, lambda n: self.grad_stage(n, self.cfg.n_blocks), bucket_bytes)
            cache[slot]["key"] = key
        return cache[slot]

    def grad_buckets(self, bucket_bytes: int = 8 << 20):
        """The flat buckets behind the parameters' ``.grad`` (list of {'flat', 'names', 'stage'}, reverse registration order)."""
        self._bucket_bytes = bucket_bytes
        return self._grad_buckets("grad", bucket_bytes)["buckets"]

    def stream_wait_grad_stage(self, stage: int, stream) -> None:
        """Make `stream` (a torch.cuda.Stream on the model's device) wait for backward stage `stage` of the last step."""
        self._use(True)
        if self._handle is None:      # (precision 'auto' before any training step: no backward has recorded a stage yet -- nothing to wait for)
            return
        capi.check(capi.lib().dinoseg_stream_wait_grad_stage(self._handle, int(stage), stream.cuda_stream))

    def _sync_grads(self, slot: str = "grad") -> dict:
        """Bind the gradient buffers of every trainable parameter to the native handle; parameters with requires_grad=False are
        unbound = frozen (freeze_bb / unfreeze_bb).  slot 'grad': the buffers ARE the parameters' ``.grad`` (views into the flat
        buckets); slot 'autograd': private buffers whose contents torch.autograd receives from DINOSeg.forward's backward."""
        lib = capi.lib()
        bk = self._grad_buckets(slot, getattr(self, "_bucket_bytes", 8 << 20))
        sig = [slot]
        for name, p in self.named_parameters():
            if p.requires_grad:
                view = bk["views"][name]
                if slot == "grad" and (p.grad is None or p.grad.data_ptr() != view.data_ptr()):
                    p.grad = view
                sig.append((name, view.data_ptr()))
            else:
                sig.append((name, None))
        sig = tuple(sig)
        if sig != self._grad_sig:
            for name, ptr in sig[1:]:
                capi.check(lib.dinoseg_bind_grad(self._handle, name.encode(), ptr))
            self._grad_sig = sig
        return bk

    def _prep_batch(self, x: torch.Tensor):
        dev = self.device
        if x.dtype == torch.uint8:
            if x.dim() != 4 or x.shape[3] != 3 or x.shape[1] != x.shape[2]:
                raise ValueError(f"expected uint8 [B,r,r,3], got {tuple(x.shape)}")
            kind, B, r = capi.INPUT_U8_HWC, x.shape[0], x.shape[1]
            x = x.to(dev).contiguous()
        else:
            if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != x.shape[3]:
                raise ValueError(f"expected [B,3,r,r], got {tuple(x.shape)}")
            kind, B, r = capi.INPUT_F32_CHW, x.shape[0], x.shape[2]
            x = x.to(device=dev, dtype=torch.float32).contiguous()
        if r % 8 != 0:
            raise ValueError("Resolution should be a multiple of 8.")
        return x, kind, B, r

    def check_labels(self) -> None:
        """Raise IndexError if a training step since the last check saw a label outside [0, n_classes) other than the
        ignore_index -100 (F.nll_loss raises at the call; here the row is skipped on device and reported on request, so the
        step stays asynchronous).  Synchronises the stream; ``fit()`` calls it once per epoch."""
        self._use(True)
        if self._handle is None:
            return
        bad = C.c_int32(0)
        capi.check(capi.lib().dinoseg_train_status(self._handle, C.byref(bad), self._stream()))
        if bad.value:
            raise IndexError(f"Target out of bounds: labels must be in [0, {self.cfg.n_classes}) or -100 (ignore_index)")

    def training_step(self, batch, batch_idx=0):
        """The reference's step verbatim (pl_torch_modules.py:261-268): ``probs = self(x); loss = F.nll_loss(probs, y)``.
        ``loss`` carries an autograd graph whose backward runs in the native library (``loss.backward()`` ACCUMULATES into
        ``.grad`` like any torch module, so zero_grad / optimizer.step around it work unchanged, Lightning included).
        ``fused_training_step`` is the same arithmetic without the autograd round trip."""
        x, y = batch
        probs = self(x)
        y = y.to(self.device).reshape((-1,)).long()
        loss = torch.nn.functional.nll_loss(probs, y)
        return {"loss": loss, "pred": probs.argmax(dim=-1).detach(), "gt": y, "probs": probs.detach()}

    def fused_training_step(self, batch, batch_idx=0):
        """zero_grad + forward + F.nll_loss + backward in one native call (``dinoseg_train_step``): on return every trainable
        parameter's ``.grad`` holds d loss / d parameter of THIS call (overwritten), so ``fused_adam_step`` -- or any torch
        optimiser -- can follow.  x: fp32 [B,3,r,r] (normalised) or uint8 [B,r,r,3]; y: int [B, (r/8)^2] (-100 = ignored)."""
        x, y = batch
        self._require_gpu()
        self._sync_weights(train=True)
        self._sync_grads("grad")
        x, kind, B, r = self._prep_batch(x)
        dev = self.device
        n = (r // 8) ** 2
        y = y.to(dev).reshape(-1).long().contiguous()
        if y.numel() != B * n:
            raise ValueError(f"labels must have B*(r/8)^2 = {B * n} entries, got {y.numel()}")
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        logp = torch.empty((B * n, self.cfg.n_classes), dtype=torch.float32, device=dev)
        capi.check(capi.lib().dinoseg_train_step(self._handle, x.data_ptr(), kind, B, r, y.data_ptr(), loss.data_ptr(),
                                                 logp.data_ptr(), self._stream()))
        self._fwd_epoch = getattr(self, "_fwd_epoch", 0) + 1
        return {"loss": loss, "pred": logp.argmax(dim=-1).detach(), "gt": y, "probs": logp}

    def _autograd_forward(self, x: torch.Tensor, kind: int, B: int, r: int) -> torch.Tensor:
        self._sync_weights(train=True)
        n = (r // 8) ** 2
        logp = torch.empty((B * n, self.cfg.n_classes), dtype=torch.float32, device=self.device)
        capi.check(capi.lib().dinoseg_train_forward(self._handle, x.data_ptr(), kind, B, r, logp.data_ptr(), self._stream()))
        self._fwd_epoch = getattr(self, "_fwd_epoch", 0) + 1
        return logp

    def _autograd_backward(self, dlogp: torch.Tensor, epoch: int, params):
        if epoch != getattr(self, "_fwd_epoch", 0):
            raise RuntimeError("DINOSeg.backward: the activations of this forward were overwritten by a later forward / "
                               "training step (one saved forward per model; call backward before the next forward)")
        self._use(True)          # (precision 'auto': an inference call may have run since the forward)
        bk = self._sync_grads("autograd")
        dlogp = dlogp.to(device=self.device, dtype=torch.float32).contiguous()
        capi.check(capi.lib().dinoseg_backward(self._handle, dlogp.data_ptr(), self._stream()))
        by_ptr = {p.data_ptr(): n for n, p in self.named_parameters()}
        # autograd may keep (not copy) what backward returns: hand out copies, the bound buffers are rewritten next time
        return tuple(bk["views"][by_ptr[p.data_ptr()]].clone() if p.requires_grad else None for p in params)

    def fused_adam_step(self, lr=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=None, grad_scale=1.0) -> None:
        """Fused Adam / AdamW update of every trainable parameter from its .grad (torch.optim semantics, bias-corrected with a
        step count kept PER PARAMETER like torch; the flavour follows self.optimizer: AdamW -> decoupled decay 0.01 by default,
        Adam -> none).  Other optimizer classes have no fused kernel: use ``configure_optimizers()`` and torch's step."""
        if self.optimizer is torch.optim.AdamW:
            decoupled = 1
        elif self.optimizer is torch.optim.Adam:
            decoupled = 0
        else:
            raise NotImplementedError(f"fused_adam_step implements torch.optim.Adam and AdamW, not {self.optimizer!r}; "
                                      "use model.configure_optimizers().step() on the .grad buffers instead")
        if weight_decay is None:
            weight_decay = 0.01 if decoupled else 0.0
        lr = self.lr if lr is None else lr
        state = self.__dict__.setdefault("_adam_state", {})
        groups = {}
        for name, p in self.named_parameters():
            if not p.requires_grad or p.grad is None:
                continue
            st = state.get(name)
            if st is None or st["m"].data_ptr() == 0 or st["m"].device != p.device:
                st = state[name] = {"m": torch.zeros_like(p), "v": torch.zeros_like(p), "step": 0}
            st["step"] += 1
            groups.setdefault(st["step"], []).append((p, st))
        for step, items in groups.items():      # one launch per distinct step count (one, unless tensors were unfrozen later)
            k = len(items)
            arr = lambda xs: (C.c_void_p * k)(*xs)
            capi.check(capi.lib().dinoseg_adam_step_multi(
                k, arr([p.data_ptr() for p, _ in items]), arr([p.grad.data_ptr() for p, _ in items]),
                arr([st["m"].data_ptr() for _, st in items]), arr([st["v"].data_ptr() for _, st in items]),
                (C.c_int64 * k)(*[p.numel() for p, _ in items]), lr, betas[0], betas[1], eps, weight_decay, decoupled, step,
                grad_scale, self._stream()))
        self.invalidate_weights()      # weights changed in place through raw pointers: re-pack on the next forward

    def training_epoch_end(self, outputs):
        """pl_torch_modules.py:343-345: the train-split metrics of an epoch (the reference computes and drops them).  Accepts the
        reference's step outputs ({'pred', 'gt', ...}) as well as this class's ({'confusion'})."""
        outs = []
        for o in outputs:
            if "confusion" not in o:
                pred = torch.as_tensor(o["pred"]).to(self.device).reshape(-1).to(torch.int32).contiguous()
                gt = torch.as_tensor(o["gt"]).to(self.device).reshape(-1).long().contiguous()
                cm = torch.zeros((self.cfg.n_classes, self.cfg.n_classes), dtype=torch.int64, device=self.device)
                capi.check(capi.lib().dinoseg_op_confusion(pred.data_ptr(), gt.data_ptr(), gt.numel(), self.cfg.n_classes,
                                                           cm.data_ptr(), self._stream()))
                o = {"confusion": cm}
            outs.append(o)
        return self.validation_epoch_end(outs, prefix="train")

    def _no_dataset(self, what):
        raise NotImplementedError(
            f"DINOSeg.{what}(): the DuckieSegDataset / albumentations pipeline (pl_torch_modules.py:347-365) is not part of "
            "dino_amd (DESIGN.md section 6); pass dataloaders to fit(), or override this hook in a subclass")

    def train_dataloader(self, sim=False):
        self._no_dataset("train_dataloader")

    def val_dataloader(self, sim=False):
        self._no_dataset("val_dataloader")

    def test_dataloader(self):
        self._no_dataset("test_dataloader")

    def _fit_phase(self, train_dataloader, val_dataloader, ck_path, max_epochs, step):
        """One ``Trainer.fit`` of the reference: ``max_epochs`` epochs, validation after each, best ``val_acc`` checkpointed.
        Every phase starts from a FRESH optimizer (moments and per-parameter step counts dropped): the reference builds a new
        ``Trainer`` per phase and per ``fit()`` call, so ``configure_optimizers()`` runs again (pl_torch_modules.py:391-421)."""
        from .ckpt import save_checkpoint
        self.__dict__.pop("_adam_state", None)
        best, history = -1.0, []
        for epoch in range(max_epochs):
            cms, losses = [], []
            for bi, (x, y) in enumerate(train_dataloader):
                out = self.fused_training_step((x, y), bi)
                self.fused_adam_step()
                losses.append(out["loss"])
                cm = torch.zeros((self.cfg.n_classes, self.cfg.n_classes), dtype=torch.int64, device=self.device)
                capi.check(capi.lib().dinoseg_op_confusion(out["pred"].to(torch.int32).contiguous().data_ptr(), out["gt"].data_ptr(),
                                                           out["gt"].numel(), self.cfg.n_classes, cm.data_ptr(), self._stream()))
                cms.append({"confusion": cm})
                step += 1
            self.check_labels()
            metrics = self.validation_epoch_end(cms, prefix="train") if cms else {}
            metrics["train_loss"] = float(torch.stack(losses).mean()) if losses else float("nan")
            metrics.update(self.validation_epoch_end([self.validation_step(b, i) for i, b in enumerate(val_dataloader)]))
            metrics["epoch"] = epoch
            history.append(metrics)
            if metrics["val_acc"] > best:
                best = metrics["val_acc"]
                save_checkpoint(self, ck_path, epoch=epoch, global_step=step)
        return history, step

    def fit(self, ck_file_name=None, train_dataloader=None, val_dataloader=None, test_dataloader=None, max_epochs=None,
            sim_dataloader=None):
        """The reference's ``fit`` (pl_torch_modules.py:367-431) without Lightning: freeze / unfreeze the backbone, train
        ``max_epochs`` epochs with ``fused_training_step`` + the fused optimizer step, validate after every epoch
        (``check_val_every_n_epoch=1``), keep the checkpoint with the best ``val_acc`` (``ModelCheckpoint(monitor='val_acc',
        mode='max')``) at ``write_path/<ck_file_name>.ckpt`` in the PL-1.5 schema, then run the test split and set
        ``self.best_ck``.  With ``pretrain_on_sim=True`` (ctor kwarg, :391-401) a first phase of ``max_epochs`` epochs runs on
        ``sim_dataloader`` (validated on the REAL validation split, like the reference's ``val_dataloader(sim=False)``) before the
        main phase; each phase tracks its own best ``val_acc`` (the reference builds a fresh ``ModelCheckpoint`` per phase), the
        main phase's best is what ``best_ck`` names.  The dataset / augmentation pipeline is out of scope,
        so the dataloaders are arguments (or the ``train_dataloader() / val_dataloader() / test_dataloader()`` hooks of a
        subclass): any iterables of ``(x, y)`` batches with x uint8 [B,r,r,3] or fp32 [B,3,r,r] and y int [B,(r/8)^2].
        Returns {'history': [per-epoch metrics of the main phase], 'sim_history': [...] or None, 'test': test metrics or None}."""
        import os

        hooks = type(self).train_dataloader is not DINOSeg.train_dataloader
        if train_dataloader is None and hooks:
            train_dataloader = self.train_dataloader()
        if val_dataloader is None and type(self).val_dataloader is not DINOSeg.val_dataloader:
            val_dataloader = self.val_dataloader()
        if test_dataloader is None and type(self).test_dataloader is not DINOSeg.test_dataloader:
            test_dataloader = self.test_dataloader()
        if train_dataloader is None or val_dataloader is None:
            raise ValueError("fit() needs train_dataloader and val_dataloader (the DuckieSegDataset pipeline is not part of dino_amd)")
        if self.pretrain_on_sim and sim_dataloader is None:
            if hooks:
                sim_dataloader = self.train_dataloader(sim=True)
            else:
                raise ValueError("pretrain_on_sim=True needs fit(sim_dataloader=...): the simulation split's loader "
                                 "(pl_torch_modules.py:391-401 builds it from train_path_sim, which is not part of dino_amd)")
        if self.freeze_backbone:
            self.freeze_bb()
        else:
            self.unfreeze_bb()
        if ck_file_name is None:        # same naming rule as the reference
            ck_file_name = (str(self.n_blocks) + "_" + self.head + ("_frozen" if self.freeze_backbone else "_finetuned") +
                            ("_grayscale" if self.grayscale else ""))
        out_dir = self.write_path if self.write_path is not None else "."
        os.makedirs(out_dir, exist_ok=True)
        ck_path = os.path.join(out_dir, ck_file_name + ".ckpt")
        epochs = self.max_epochs if max_epochs is None else max_epochs
        sim_history, step = None, 0
        if self.pretrain_on_sim:
            sim_history, step = self._fit_phase(sim_dataloader, val_dataloader, ck_path, epochs, 0)
        history, step = self._fit_phase(train_dataloader, val_dataloader, ck_path, epochs, 0)
        self.best_ck = ck_path if history else None
        test = None
        if test_dataloader is not None:
            test = self.test_epoch_end([self.test_step(b, i) for i, b in enumerate(test_dataloader)])
        if self.comet_logger is not None and self.best_ck is not None:
            self.comet_logger.experiment.log_asset(self.best_ck)
        return {"history": history, "sim_history": sim_history, "test": test}

    def freeze_bb(self):
        for p in self.dino.parameters():
            p.requires_grad = False

    def unfreeze_bb(self):
        for p in self.dino.parameters():
            p.requires_grad = True

    def configure_optimizers(self):
        return self.optimizer(self.parameters(), lr=self.lr)

    # ---- checkpoints ------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **overrides):
        from .ckpt import load_checkpoint
        return load_checkpoint(cls, checkpoint_path, map_location=map_location, **overrides)
