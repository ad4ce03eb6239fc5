"""hcir.resize — Pillow-exact bicubic Resize + CenterCrop on the HIP device (include/hcir.h, csrc/resize.hip).

Stands where the hair_retrieval path resizes on the host: `transforms.Resize(224, interpolation=3)` ->
`CenterCrop(224)` (src/models/hair_encoder.py:44-48; applied at :116-117 and :175), i.e. torchvision ->
`PIL.Image.resize((ow, oh), BICUBIC)`.  The coefficient tables are Pillow's (computed on the host in double by
`hcir_resize_bicubic_coeffs`, cached per size pair); the two passes run on the device in Pillow's integer
arithmetic, only for the pixels the crop window needs.  There is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from ._lib import HcirError, check


class ResizeJob(ctypes.Structure):
    """Mirror of hcir_resize_job (include/hcir.h)."""
    _fields_ = [("src_offset", ctypes.c_uint64), ("src_pitch", ctypes.c_int64), ("src_h", ctypes.c_int32),
                ("src_w", ctypes.c_int32), ("out_h", ctypes.c_int32), ("out_w", ctypes.c_int32),
                ("crop_top", ctypes.c_int32), ("crop_left", ctypes.c_int32), ("coef_h", ctypes.c_int32),
                ("coef_v", ctypes.c_int32), ("ksize_h", ctypes.c_int32), ("ksize_v", ctypes.c_int32)]


JOB_BYTES = ctypes.sizeof(ResizeJob)
_axis_cache: Dict[Tuple[int, int], Tuple[int, np.ndarray]] = {}


def axis_table(in_size: int, out_size: int) -> Tuple[int, np.ndarray]:
    """(ksize, [bounds (2 * out) | kk (out * ksize)] int32) of one axis, Pillow's tables (host, cached)."""
    key = (int(in_size), int(out_size))
    hit = _axis_cache.get(key)
    if hit is None:
        L = _lib.lib()
        ks = L.hcir_resize_bicubic_ksize(*key)
        if ks <= 0:
            raise HcirError(f"bad resize {key}")
        tab = np.zeros(2 * key[1] + key[1] * ks, np.int32)
        check(L.hcir_resize_bicubic_coeffs(key[0], key[1], tab.ctypes.data, tab[2 * key[1]:].ctypes.data),
              "hcir_resize_bicubic_coeffs")
        hit = _axis_cache[key] = (ks, tab)
    return hit


def resize_output_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision transforms.Resize(int) on a PIL image: shorter side -> size, the other int(size * long / short);
    an image whose shorter side already is `size` is returned as it is (src/models/hair_encoder.py:46)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_origin(oh: int, ow: int, win_h: int, win_w: int) -> Tuple[int, int]:
    """Origin of torchvision's CenterCrop window inside an oh x ow image (negative: it pads with zeros first)."""
    ph, pw = max(win_h - oh, 0), max(win_w - ow, 0)
    top = int(round((oh + ph - win_h) / 2.0)) - ph // 2
    left = int(round((ow + pw - win_w) / 2.0)) - pw // 2
    return top, left


_ws = {}


def resize_center_crop(images: Union[torch.Tensor, Sequence[torch.Tensor]], size: int = 224,
                       out_sizes: Sequence[Tuple[int, int]] = None) -> torch.Tensor:
    """Resize(size, bicubic) -> CenterCrop(size) of RGB8 device images: one [b, h, w, 3] tensor or a list of
    [h, w, 3] tensors of any sizes -> uint8 [b, size, size, 3].  out_sizes: (oh, ow) per image instead of the
    shorter-side rule.  Asynchronous on the current stream."""
    imgs: List[torch.Tensor] = list(images) if not isinstance(images, torch.Tensor) else list(images.unbind(0))
    if not imgs:
        raise HcirError("resize_center_crop needs at least one image")
    dev = imgs[0].device
    for t in imgs:
        if not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 3 or t.size(2) != 3 or t.stride(2) != 1 or \
                t.stride(1) != 3 or t.device != dev:
            raise HcirError("resize_center_crop takes RGB8 [h, w, 3] tensors (pixel-contiguous) on one HIP device "
                            "(no CPU fallback)")
    L = _lib.lib()
    b = len(imgs)
    base = min(t.data_ptr() for t in imgs)
    tables: List[np.ndarray] = []
    offs: Dict[Tuple[int, int], int] = {}
    total = 0

    def table_offset(i, o):
        nonlocal total
        if i == o:
            return -1, 0
        key = (i, o)
        ks, tab = axis_table(i, o)
        if key not in offs:
            offs[key] = total
            tables.append(tab)
            total += tab.size
        return offs[key], ks

    jobs = (ResizeJob * b)()
    for n, t in enumerate(imgs):
        h, w = int(t.size(0)), int(t.size(1))
        oh, ow = out_sizes[n] if out_sizes is not None else resize_output_size(h, w, size)
        top, left = center_origin(oh, ow, size, size)
        ch, kh = table_offset(w, ow)
        cv, kv = table_offset(h, oh)
        jobs[n] = ResizeJob(t.data_ptr() - base, t.stride(0), h, w, oh, ow, top, left, ch, cv, kh, kv)
    host = torch.empty(b * JOB_BYTES + 4 * max(total, 1), dtype=torch.uint8, pin_memory=True)
    ctypes.memmove(host.data_ptr(), ctypes.addressof(jobs), b * JOB_BYTES)
    if total:
        coef = np.concatenate(tables)
        ctypes.memmove(host.data_ptr() + b * JOB_BYTES, coef.ctypes.data, coef.nbytes)
    blob = host.to(dev, non_blocking=True)
    out = torch.empty((b, size, size, 3), dtype=torch.uint8, device=dev)
    wsb = L.hcir_resize_crop_workspace_bytes(ctypes.addressof(jobs), b, size, size)
    stream = torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < wsb:
        ws = _ws[key] = torch.empty(wsb, dtype=torch.uint8, device=dev)
    check(L.hcir_resize_crop_bicubic_u8(base, blob.data_ptr() + b * JOB_BYTES, blob.data_ptr(), ctypes.addressof(jobs),
                                        b, size, size, out.data_ptr(), ws.data_ptr(), ws.numel(), stream),
          "hcir_resize_crop_bicubic_u8")
    out._hcir_keepalive = (blob, host, imgs)  # the kernels read these after this call returns
    return out
