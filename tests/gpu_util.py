"""Helpers for the -m gpu parity tests: everything goes through the C-ABI (dino_amd.capi)."""
from __future__ import annotations

import numpy as np
import torch

from dino_amd import capi


def pack(x: torch.Tensor, planes: int, rows_pad: int = None, cols_pad: int = None) -> torch.Tensor:
    """fp32 [rows, cols] device tensor -> int16 view of bf16 planes [planes, rows_pad, cols_pad]."""
    assert x.dtype == torch.float32 and x.is_cuda and x.dim() == 2
    x = x.contiguous()
    rows, cols = x.shape
    rows_pad = rows_pad or rows
    cols_pad = cols_pad or cols
    out = torch.empty((planes, rows_pad, cols_pad), dtype=torch.int16, device=x.device)
    capi.check(capi.lib().dinoseg_op_pack(x.data_ptr(), rows, cols, out.data_ptr(), rows_pad * cols_pad, rows_pad,
                                          cols_pad, planes, capi.stream_ptr()))
    return out


def unpack(p: torch.Tensor) -> torch.Tensor:
    """int16 bf16 planes [planes, ...] -> fp32 sum of the planes."""
    return p.view(torch.bfloat16).to(torch.float32).sum(dim=0)


def quant_like(x: torch.Tensor, planes: int) -> torch.Tensor:
    """What the kernels see of an fp32 operand: bf16(x) or bf16(x) + bf16(x - bf16(x))."""
    hi = x.to(torch.bfloat16).to(torch.float32)
    if planes == 1:
        return hi
    return hi + (x - hi).to(torch.bfloat16).to(torch.float32)


def seeded(shape, seed, scale=1.0, device="cuda"):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32)).to(device)


def pack_slabs(W: torch.Tensor, planes: int) -> torch.Tensor:
    """fp32 [N, K] device weight -> int16 view of the slab-major bf16 copy gemm_ln.hip streams."""
    N, K = W.shape
    n = capi.lib().dinoseg_op_ln_gemm_slab_elems(N, K, planes)
    assert n > 0
    out = torch.empty((n,), dtype=torch.int16, device=W.device)
    capi.check(capi.lib().dinoseg_op_pack_slabs(W.contiguous().data_ptr(), N, K, planes, out.data_ptr(), capi.stream_ptr()))
    return out
