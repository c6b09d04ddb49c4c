"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 2): scalar restatement of cv2.resize(INTER_LINEAR) on uint8.

What it restates: the Resize(r, r) step of ``get_transforms`` (dt_segmentation/src/pl_torch_modules.py:36-38, applied in
``predict`` at :291) = albumentations 1.1.0 -> ``cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)``.
OpenCV (opencv_python==4.5.5.62, dt_segmentation/requirements.txt:7) is third-party, not vendored in /root/reference and
not installed in this image, and the reference holds no golden image for it: PARITY UNPINNED.  The algorithm below is the
published one of modules/imgproc/src/resize.cpp (resizeGeneric_ with HResizeLinear<uchar,int,short,...> and
VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>, and the INTER_LINEAR -> INTER_AREA switch at exactly 2x),
written one pixel at a time, independently of dino_amd/preprocess.py and of the HIP kernel, which tests compare to it.
"""
from __future__ import annotations

import math
import struct

import numpy as np


def _f32(x: float) -> float:
    return struct.unpack("f", struct.pack("f", x))[0]


def _coef(v: float) -> int:
    """saturate_cast<short>(v * INTER_RESIZE_COEF_SCALE): float product, round half to even, clamp to int16."""
    r = round(_f32(_f32(v) * 2048.0))          # Python's round() on a float is round-half-even like cvRound / lrintf
    return max(-32768, min(32767, r))


def resize_linear_u8(img: np.ndarray, dh: int, dw: int) -> np.ndarray:
    sh, sw, ch = img.shape
    out = np.zeros((dh, dw, ch), dtype=np.uint8)
    if sw == 2 * dw and sh == 2 * dh:
        for y in range(dh):
            for x in range(dw):
                for c in range(ch):
                    out[y, x, c] = (int(img[2 * y, 2 * x, c]) + int(img[2 * y, 2 * x + 1, c]) + int(img[2 * y + 1, 2 * x, c]) +
                                    int(img[2 * y + 1, 2 * x + 1, c]) + 2) >> 2
        return out
    scale_x = 1.0 / (dw / sw)
    scale_y = 1.0 / (dh / sh)
    xs = []
    for dx in range(dw):
        fx = _f32((dx + 0.5) * scale_x - 0.5)
        sx = math.floor(fx)
        fx = _f32(fx - sx)
        if sx < 0:
            fx, sx = 0.0, 0
        if sx >= sw - 1:
            fx, sx = 0.0, sw - 1
        xs.append((sx, min(sx + 1, sw - 1), _coef(_f32(1.0 - fx)), _coef(fx)))
    for dy in range(dh):
        fy = _f32((dy + 0.5) * scale_y - 0.5)
        sy = math.floor(fy)
        fy = _f32(fy - sy)
        y0 = min(max(sy, 0), sh - 1)
        y1 = min(max(sy + 1, 0), sh - 1)
        b0, b1 = _coef(_f32(1.0 - fy)), _coef(fy)
        for dx, (x0, x1, a0, a1) in enumerate(xs):
            for c in range(ch):
                h0 = int(img[y0, x0, c]) * a0 + int(img[y0, x1, c]) * a1
                h1 = int(img[y1, x0, c]) * a0 + int(img[y1, x1, c]) * a1
                out[dy, dx, c] = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xFF
    return out
