"""ctypes binding of ``libdinoseg_hip.so`` (C-ABI declared in ``include/dinoseg.h``).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is missing
or does not export a declared symbol, importing/using this module raises immediately.
PyTorch is used only as the owner of device memory and streams (``tensor.data_ptr()``,
``torch.cuda.current_stream().cuda_stream``).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

import torch  # noqa: F401  (must be imported first: the library binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdinoseg_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dinoseg.h")

BF16, BF16X3, FP16, FP16X3 = 0, 1, 2, 3
HEAD_LINEAR, HEAD_MLP = 0, 1
INPUT_U8_HWC, INPUT_F32_CHW = 0, 1
EPI_PLAIN, EPI_RESID, EPI_GELU, EPI_RELU = 0, 1, 2, 3
PROF_CLASSES = ("patch_embed", "layernorm", "qkv_gemm", "attention", "proj_gemm", "fc1_gemm", "fc2_gemm", "head", "attention_bwd")


class DinosegError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("embed_dim", C.c_int32), ("num_heads", C.c_int32), ("n_blocks", C.c_int32), ("patch", C.c_int32),
        ("mlp_ratio", C.c_int32), ("n_classes", C.c_int32), ("head_kind", C.c_int32), ("pos_grid", C.c_int32),
        ("ln_eps", C.c_float), ("precision", C.c_int32),
    ]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_fp = C.c_void_p   # device float* / int32_t* travel as raw addresses

# name -> (restype, argtypes); must list every function include/dinoseg.h declares (checked by tests)
SIGNATURES = {
    "dinoseg_last_error": (C.c_char_p, []),
    "dinoseg_version": (C.c_int, []),
    "dinoseg_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "dinoseg_destroy": (C.c_int, [_vp]),
    "dinoseg_bind_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32]),
    "dinoseg_refresh_weights": (C.c_int, [_vp, _vp]),
    "dinoseg_prepare_resolution": (C.c_int, [_vp, _i32, _vp]),
    "dinoseg_forward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _fp, _fp, _i32, _fp, _vp]),
    "dinoseg_workspace_bytes": (_i64, [_vp, _i32, _i32]),
    "dinoseg_state_generation": (_i64, [_vp]),
    "dinoseg_last_selfattention": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _fp, _vp]),
    "dinoseg_op_confusion": (C.c_int, [_fp, _fp, _i64, _i32, _fp, _vp]),
    "dinoseg_forward_mask": (C.c_int, [_vp, _vp, _i32, _i32, _fp, _i32, _fp, _fp, _vp]),
    "dinoseg_features": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _fp, _vp]),
    "dinoseg_train_forward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _fp, _vp]),
    "dinoseg_backward": (C.c_int, [_vp, _fp, _vp]),
    "dinoseg_grad_stages": (C.c_int, [_vp]),
    "dinoseg_stream_wait_grad_stage": (C.c_int, [_vp, _i32, _vp]),
    "dinoseg_train_status": (C.c_int, [_vp, C.POINTER(_i32), _vp]),
    "dinoseg_op_resize_u8": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _i32, _vp]),
    "dinoseg_bind_grad": (C.c_int, [_vp, C.c_char_p, _fp]),
    "dinoseg_train_step": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _fp, _fp, _fp, _vp]),
    "dinoseg_adam_step": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _i32, _f32, _vp]),
    "dinoseg_adam_step_multi": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32,
                                           _i32, _i32, _f32, _vp]),
    "dinoseg_op_attention_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _fp, _fp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dinoseg_op_layernorm_bwd": (C.c_int, [_fp, _fp, _fp, _f32, _i32, _i32, _fp, _i32, _fp, _fp, _i32, _i32, _vp]),
    "dinoseg_set_option": (C.c_int, [C.c_char_p, _i32]),
    "dinoseg_profile": (C.c_int, [_vp, _i32]),
    "dinoseg_profile_read": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(_i32)]),
    "dinoseg_op_pack": (C.c_int, [_fp, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _vp]),
    "dinoseg_op_gemm": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _fp, _fp, _vp, _i64, _i32, _vp]),
    "dinoseg_op_qkv_gemm": (C.c_int, [_vp, _i64, _vp, _i64, _fp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i64, _vp]),
    "dinoseg_op_ln_gemm_slab_elems": (_i64, [_i32, _i32, _i32]),
    "dinoseg_op_pack_slabs": (C.c_int, [_fp, _i32, _i32, _i32, _vp, _vp]),
    "dinoseg_op_ln_gemm": (C.c_int, [_fp, _fp, _fp, _f32, _vp, _i64, _fp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64,
                                     _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "dinoseg_op_mlp_fused_pack_elems": (_i64, [_i32, _i32]),
    "dinoseg_op_pack_mlp": (C.c_int, [_fp, _fp, _i32, _i32, _vp, _vp]),
    "dinoseg_op_mlp_fused": (C.c_int, [_fp, _fp, _fp, _f32, _vp, _fp, _fp, _i32, _i32, _i32, _vp]),
    "dinoseg_op_qkv_pack_elems": (_i64, [_i32]),
    "dinoseg_op_pack_qkv": (C.c_int, [_fp, _i32, _vp, _vp]),
    "dinoseg_op_block_tail_fused": (C.c_int, [_fp, _vp, _vp, _fp, _fp, _fp, _f32, _vp, _fp, _fp, _vp, _fp, _fp, _fp, _vp, _vp, _vp, _i32,
                                              _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "dinoseg_op_proj_pack_elems": (_i64, [_i32]),
    "dinoseg_op_pack_proj": (C.c_int, [_fp, _i32, _vp, _vp]),
    "dinoseg_op_proj_mlp_fused": (C.c_int, [_fp, _vp, _vp, _fp, _fp, _fp, _f32, _vp, _fp, _fp, _i32, _i32, _i32, _vp]),
    "dinoseg_op_pack_rs": (C.c_int, [_fp, _i32, _i32, _i32, _vp, _vp]),
    "dinoseg_op_gemm_rs": (C.c_int, [_vp, _i32, _vp, _fp, _i32, _i32, _i32, _i32, _fp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "dinoseg_op_pack_rs_ln": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _vp, _fp, _vp]),
    "dinoseg_op_ln_gemm_rs": (C.c_int, [_fp, _f32, _vp, _fp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "dinoseg_op_mlp3_pack_elems": (_i64, [_i32, _i32]),
    "dinoseg_op_pack_mlp3": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _vp, _vp]),
    "dinoseg_op_proj_mlp_fused3": (C.c_int, [_fp, _vp, _i64, _fp, _f32, _vp, _fp, _i32, _i32, _i32, _i32, _vp]),
    "dinoseg_op_mlp4_pack_elems": (_i64, [_i32, _i32]),
    "dinoseg_op_pack_mlp4": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _vp, _vp]),
    "dinoseg_op_proj_mlp_fused4": (C.c_int, [_fp, _vp, _fp, _f32, _vp, _fp, _i32, _i32, _i32, _i32, _vp]),
    "dinoseg_op_block_tail_fused4": (C.c_int, [_fp, _vp, _fp, _f32, _vp, _fp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "dinoseg_op_block_tail_fused3": (C.c_int, [_fp, _vp, _i64, _fp, _f32, _vp, _fp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _i32,
                                               _i32, _i32, _vp]),
    "dinoseg_op_attention": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _fp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dinoseg_op_layernorm": (C.c_int, [_fp, _fp, _fp, _f32, _i32, _i32, _vp, _i64, _i32, _fp, _i32, _i32, _vp]),
    "dinoseg_op_pos_resample": (C.c_int, [_fp, _i32, _i32, _i32, _fp, _vp]),
    "dinoseg_op_patch_gather": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "dinoseg_op_head_final": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _fp, _fp, _i32, _fp, _fp, _vp]),
}

_lib: Optional[C.CDLL] = None


def header_symbols() -> list:
    """Function names declared in include/dinoseg.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(dinoseg_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    """Load the shared library once; raise loudly if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("DINOSEG_LIB", LIB_PATH)      # A/B runs of two builds (tools/ab_bench.sh); default: the in-tree build
    if not os.path.exists(path):
        raise DinosegError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C dino_amd/csrc`). dino_amd has no CPU/PyTorch fallback.")
    l = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def last_error() -> str:
    return lib().dinoseg_last_error().decode()


def check(rc: int) -> None:
    if rc != 0:
        msg = last_error()
        # the reference raises ValueError for a bad resolution (pl_torch_modules.py:271-272)
        if msg == "Resolution should be a multiple of 8.":
            raise ValueError(msg)
        raise DinosegError(f"dinoseg error {rc}: {msg}")


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
