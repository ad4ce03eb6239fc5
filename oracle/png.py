"""oracle.png — TEST INFRASTRUCTURE ONLY (never imported by the product path).

ctypes face of oracle/png_oracle.c: the CPU restatement of the PNG decode the reference's loader runs before
`knn_transform` (`HairPretraining/utils/dataloader.py:28-31`: `read_file` -> `torchvision.io.decode_image(..., RGB)`;
`src/models/hair_encoder.py:108,169`: `PIL.Image.open(...).convert('RGB')`).  The arithmetic lives in third-party
libraries that are not under /root/reference (libpng / Pillow's PngImagePlugin, zlib); what is restated is their
published format: PNG chunks + scanline filters (PNG spec §5, §9), zlib (RFC 1950), inflate (RFC 1951).
PIN: tests/test_png_host.py — byte for byte against `zlib.decompress` and Pillow 12.2 on the four
`assets/hair_region_only/*.png` goldens and on seeded synthetic files (every filter type, every block type).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s CPU-baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

STAT_NAMES = ("stored_blocks", "fixed_blocks", "dynamic_blocks", "literals", "matches", "match_bytes", "overlapping",
              "dist_le_4k", "dist_le_8k", "dist_le_16k", "dist_gt_16k", "long_codes", "consumed_bytes", "codes_gt_11",
              "codes_gt_12")


class Corrupt(ValueError):
    pass


class Unsupported(ValueError):
    """Outside the 8-bit non-interlaced subset."""


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(os.path.join(_HERE, "libpng_oracle.so"))
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.png_oracle_inflate.restype = ctypes.c_int
        L.png_oracle_inflate.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), vp]
        L.png_oracle_parse.restype = ctypes.c_int
        L.png_oracle_parse.argtypes = [vp, sz, vp, vp, ctypes.POINTER(sz), vp, ctypes.c_int]
        L.png_oracle_unfilter_rgb.restype = ctypes.c_int
        L.png_oracle_unfilter_rgb.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
        _lib = L
    return _lib


def inflate(stream: bytes, cap: int, stats: bool = False):
    """First `cap` bytes of the zlib stream's output (all of it, Adler-32 verified, when it is shorter)."""
    src = np.frombuffer(stream, np.uint8)
    out = np.zeros(max(cap, 1), np.uint8)
    st = np.zeros(16, np.uint64)
    n = ctypes.c_size_t(0)
    rc = lib().png_oracle_inflate(src.ctypes.data, src.size, out.ctypes.data, cap, ctypes.byref(n), st.ctypes.data)
    if rc < 0:
        raise Corrupt(f"inflate failed ({rc}) after {n.value} bytes")
    res = out[:n.value].tobytes()
    return (res, dict(zip(STAT_NAMES, (int(v) for v in st)))) if stats else res


def parse(file: bytes, check_crc: bool = True) -> Dict:
    src = np.frombuffer(file, np.uint8)
    info = np.zeros(5, np.int32)
    idat = np.zeros(src.size, np.uint8)
    pal = np.zeros(768, np.uint8)
    n = ctypes.c_size_t(0)
    rc = lib().png_oracle_parse(src.ctypes.data, src.size, info.ctypes.data, idat.ctypes.data, ctypes.byref(n),
                                pal.ctypes.data, int(check_crc))
    if rc == -3:
        raise Unsupported("not a PNG")
    if rc < 0:
        raise Corrupt(f"chunk walk failed ({rc})")
    w, h, depth, ctype, interlace = (int(v) for v in info)
    return {"width": w, "height": h, "depth": depth, "color_type": ctype, "interlace": interlace,
            "idat": idat[:n.value].tobytes(), "palette": pal}


def bytes_per_pixel(color_type: int) -> int:
    return {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]


def decode(file: bytes, rows: Optional[int] = None) -> np.ndarray:
    """RGB8 [rows or height, width, 3] as `Image.open(...).convert("RGB")` gives it (rows: only the first rows)."""
    p = parse(file)
    if p["depth"] != 8 or p["interlace"] or p["color_type"] not in (0, 2, 3, 4, 6):
        raise Unsupported(f"depth {p['depth']} interlace {p['interlace']} colour type {p['color_type']}")
    w, h = p["width"], p["height"]
    rows = h if rows is None else min(rows, h)
    stride = 1 + w * bytes_per_pixel(p["color_type"])
    raw = inflate(p["idat"], rows * stride)
    if len(raw) != rows * stride:
        raise Corrupt("the stream ends before the last row")
    buf = np.frombuffer(raw, np.uint8).copy()
    rgb = np.zeros((rows, w, 3), np.uint8)
    rc = lib().png_oracle_unfilter_rgb(buf.ctypes.data, w, rows, p["color_type"], p["palette"].ctypes.data,
                                       rgb.ctypes.data)
    if rc < 0:
        raise Corrupt("filter type > 4")
    return rgb


def center_window(rgb: np.ndarray, wh: int, ww: Optional[int] = None) -> np.ndarray:
    """torchvision CenterCrop: pad with zeros when the image is smaller, then crop (HP/utils/transform.py:11)."""
    ww = wh if ww is None else ww
    h, w = rgb.shape[:2]
    ph, pw = max(wh - h, 0), max(ww - w, 0)
    if ph or pw:
        rgb = np.pad(rgb, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = rgb.shape[:2]
    top, left = int(round((h - wh) / 2.0)), int(round((w - ww) / 2.0))
    return rgb[top:top + wh, left:left + ww]


def window_rows(h: int, wh: int) -> Tuple[int, int]:
    """Image rows [y0, y1) a CenterCrop(wh) window of an h-row image touches."""
    if h <= wh:
        return 0, h
    top = int(round((h - wh) / 2.0))
    return top, top + wh
