"""PyTorch-Lightning-free reader/writer for the reference's ``.ckpt`` files.

The reference relies on ``LightningModule.load_from_checkpoint`` (PL 1.5.10; call sites README.md:31,
visualize.py:23, run_experiment.py:116): ``torch.load`` -> ``cls(**ckpt['hyper_parameters'])`` ->
strict ``load_state_dict(ckpt['state_dict'])``.  PL checkpoints pickle arbitrary ctor arguments
(``optimizer=<class AdamW>``, a Comet logger instance, ...; pl_torch_modules.py:145-147,225), so the
unpickler below substitutes an inert placeholder for any class that cannot be imported here.
No reference checkpoint exists in-tree (README.md:9: Google Drive), so the schema is pinned only by
the fixture this module writes itself (parity unpinned at this boundary).
"""
from __future__ import annotations

import inspect
import pickle
import types
from collections import OrderedDict

import torch

_CTOR_KEYS = None


class _Placeholder:
    """Stands in for an un-importable pickled class (e.g. comet / pytorch_lightning objects)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):
        return self


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            return type(name, (_Placeholder,), {"__module__": module})


_tolerant_pickle = types.SimpleNamespace(
    Unpickler=_TolerantUnpickler, load=lambda f, **kw: _TolerantUnpickler(f, **kw).load(),
    __name__="dino_amd_tolerant_pickle")


def read_checkpoint(path, map_location=None) -> dict:
    return torch.load(path, map_location=map_location or "cpu", pickle_module=_tolerant_pickle, weights_only=False)


def load_checkpoint(cls, path, map_location=None, **overrides):
    ck = read_checkpoint(path, map_location)
    if "state_dict" not in ck:
        raise KeyError(f"{path}: not a Lightning-style checkpoint (no 'state_dict')")
    hp = dict(ck.get("hyper_parameters", {}) or {})
    allowed = set(inspect.signature(cls.__init__).parameters) - {"self"}
    kwargs = {}
    for k, v in hp.items():
        if k not in allowed:
            continue
        if isinstance(v, _Placeholder) or (isinstance(v, type) and issubclass(v, _Placeholder)):
            continue
        kwargs[k] = v
    kwargs.update(overrides)
    sd = ck["state_dict"]
    if "n_blocks" not in kwargs:   # infer the architecture from the tensors if hyper-parameters are absent
        n = 0
        while f"dino.blocks.{n}.norm1.weight" in sd:
            n += 1
        kwargs["n_blocks"] = n
    if "head" not in kwargs:
        kwargs["head"] = "mlp" if "clf.layer_2.weight" in sd else "linear"
    if "arch" not in kwargs and "dino.cls_token" in sd:
        kwargs["arch"] = "vit_base" if sd["dino.cls_token"].shape[-1] == 768 else "vit_small"
    model = cls(**kwargs)
    model.load_state_dict(OrderedDict((k, v.to(torch.float32)) for k, v in sd.items()), strict=True)
    return model


def save_checkpoint(model, path, epoch: int = 0, global_step: int = 0) -> None:
    """Write a PL-1.5-shaped checkpoint the reference's ``load_from_checkpoint`` schema expects."""
    hp = {k: getattr(model, k) for k in (
        "class_names", "head", "n_blocks", "batch_size", "lr", "optimizer", "freeze_backbone", "max_epochs",
        "patience", "grayscale", "n_classes", "pretrain_on_sim", "augmented", "random_init", "backbone")}
    hp["data_path"], hp["write_path"], hp["comet_logger"] = model.data_path, model.write_path, None
    torch.save({
        "epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "1.5.10",
        "state_dict": OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items()),
        "hyper_parameters": hp,
    }, path)
