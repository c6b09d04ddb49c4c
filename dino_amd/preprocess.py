"""Host-side mirror of the reference's resize step (``A.Resize(r, r)`` in ``get_transforms``, pl_torch_modules.py:36-38).

albumentations 1.1.0 calls ``cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)`` on the uint8 frame.  OpenCV
(opencv_python==4.5.5.62, requirements.txt:7) is a third-party dependency that is not part of the reference tree and is
not installed here, so this restates its published fixed-point algorithm (modules/imgproc/src/resize.cpp); the same
arithmetic runs on the GPU in ``dinoseg_op_resize_u8`` (csrc/elementwise.hip), which is what ``DINOSeg.predict`` uses.
Parity against cv2 itself is unpinned; the two implementations and oracle/resize_oracle.py are
checked against each other bit for bit.
"""
from __future__ import annotations

import numpy as np

_COEF_SCALE = np.float32(2048.0)          # INTER_RESIZE_COEF_SCALE = 1 << 11


def _axis_taps(dst: int, src: int, zero_frac_when_clamped: bool):
    """Source index pair and int coefficients of one axis (float32 / float64 steps exactly as cv::resize takes them)."""
    scale = 1.0 / (float(dst) / float(src))                                     # double: scale = 1. / inv_scale
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if zero_frac_when_clamped:                                                   # x axis: clamped taps get weight (1, 0)
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
        s0, s1 = s, np.minimum(s + 1, src - 1)
    else:                                                                        # y axis: rows are clamped, weights kept
        s0, s1 = np.clip(s, 0, src - 1), np.clip(s + 1, 0, src - 1)
    c0 = np.clip(np.rint((np.float32(1) - f) * _COEF_SCALE), -32768, 32767).astype(np.int32)
    c1 = np.clip(np.rint(f * _COEF_SCALE), -32768, 32767).astype(np.int32)
    return s0, s1, c0, c1


def resize_linear_u8(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    """uint8 [H, W, C] -> uint8 [dh, dw, C], cv2.INTER_LINEAR arithmetic (see module docstring)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim != 3:
        raise ValueError(f"expected an HxWxC image, got {img.shape}")
    sh, sw = img.shape[:2]
    if (sh, sw) == (dh, dw):
        return img
    if sw == 2 * dw and sh == 2 * dh:        # cv::resize switches INTER_LINEAR to the INTER_AREA fast path at exactly 2x
        a = img.astype(np.int32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _axis_taps(dw, sw, True)
    y0, y1, b0, b1 = _axis_taps(dh, sh, False)
    src = img.astype(np.int32)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]       # horizontal pass, int32
    h0, h1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((b0[:, None, None] * h0) >> 16) + ((b1[:, None, None] * h1) >> 16) + 2) >> 2
    return out.astype(np.uint8)
