"""oracle.resize — TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the resize in front of the hair_retrieval path's model:
`transforms.Resize(224, interpolation=3)` (src/models/hair_encoder.py:46) -> torchvision `F.resize` on a PIL image ->
`PIL.Image.resize((ow, oh), BICUBIC)`.  The arithmetic lives in Pillow (third-party, not under /root/reference):
libImaging/Resample.c — `precompute_coeffs`, `normalize_coeffs_8bpc` (PRECISION_BITS = 22),
`ImagingResampleHorizontal_8bpc`, `ImagingResampleVertical_8bpc`, bicubic kernel a = -0.5, horizontal pass first
with an 8-bit clipped intermediate.  Restated here in plain Python floats (IEEE doubles, the same operation order)
and numpy integer arithmetic.
PIN: tests/test_resize_host.py — byte for byte against live Pillow 12.2 `Image.resize(..., BICUBIC)` over a sweep of
up- and down-scales, odd sizes, 1024^2 -> 224^2.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s CPU-baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(ksize, bounds [out, 2], kk [out, ksize] int32): precompute_coeffs + normalize_coeffs_8bpc for the whole axis."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img: np.ndarray, out_size: int) -> np.ndarray:
    """Resample axis 1 of img [n, in, c] uint8 to out_size."""
    n, in_size, c = img.shape
    if in_size == out_size:
        return img
    _, bounds, kk = coeffs(in_size, out_size)
    out = np.zeros((n, out_size, c), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, xmin:xmin + cnt, :], kk[xx, :cnt].astype(np.int64), ([1], [0]))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize(rgb: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """Image.resize((ow, oh), BICUBIC) of an RGB8 array [h, w, 3]: horizontal pass, then vertical."""
    t = _pass(rgb, ow)
    return _pass(t.transpose(1, 0, 2), oh).transpose(1, 0, 2)


def resize_output_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): the shorter side becomes `size`, the other int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)
