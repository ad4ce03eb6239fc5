"""A tiny PNG writer for the PNG tests and goldens.  Test infrastructure.

It exists to FORCE what an off-the-shelf encoder chooses by heuristics: the filter type of every scanline, the
deflate block type (stored / fixed / dynamic, and a mix of them in one stream with matches reaching across the
seams), the IDAT chunking.  Compression itself is zlib's; what is written by hand is the PNG container (signature,
chunks, CRC-32), the five scanline filters (PNG spec §9.2) and the zlib wrapper around hand-joined raw deflate
segments (RFC 1950: CMF/FLG + Adler-32).
"""
from __future__ import annotations

import struct
import zlib
from typing import List, Optional, Sequence, Union

import numpy as np

CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _paeth(a, b, c):
    a, b, c = a.astype(np.int32), b.astype(np.int32), c.astype(np.int32)
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def filter_rows(img: np.ndarray, filters: Sequence[int]) -> bytes:
    """img [h, w, c] uint8 -> filtered scanlines (filter byte + row), filter type per row as given."""
    h, w, c = img.shape
    flat = img.reshape(h, w * c).astype(np.int32)
    out = np.zeros((h, 1 + w * c), np.uint8)
    zero = np.zeros(w * c, np.int32)
    for y in range(h):
        cur, up = flat[y], flat[y - 1] if y else zero
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]]) if w * c > c else np.zeros(w * c, np.int32)
        upleft = np.concatenate([np.zeros(c, np.int32), up[:-c]]) if w * c > c else np.zeros(w * c, np.int32)
        ft = int(filters[y])
        pred = (zero, left, up, (left + up) >> 1, _paeth(left, up, upleft))[ft]
        out[y, 0] = ft
        out[y, 1:] = ((cur - pred) & 255).astype(np.uint8)
    return out.tobytes()


def deflate_segments(data: bytes, plan: Sequence[tuple]) -> bytes:
    """zlib stream whose deflate blocks follow `plan`: a list of (nbytes | None, kind, level) with kind in
    {"stored", "fixed", "dynamic", "huffman", "rle"}.  Every segment is a raw deflate run that sees the previous
    32 KB as its dictionary (so matches cross the seams); all but the last end on a sync flush."""
    raw = bytearray()
    pos = 0
    for i, (n, kind, level) in enumerate(plan):
        lastseg = i == len(plan) - 1
        n = len(data) - pos if (n is None or lastseg) else min(n, len(data) - pos)
        strategy = {"stored": zlib.Z_DEFAULT_STRATEGY, "fixed": zlib.Z_FIXED, "dynamic": zlib.Z_DEFAULT_STRATEGY,
                    "huffman": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE}[kind]
        lvl = 0 if kind == "stored" else level
        zd = bytes(data[max(0, pos - 32768):pos])
        co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strategy, zd) if zd else \
            zlib.compressobj(lvl, zlib.DEFLATED, -15, 9, strategy)
        raw += co.compress(bytes(data[pos:pos + n]))
        raw += co.flush(zlib.Z_FINISH if lastseg else zlib.Z_SYNC_FLUSH)
        pos += n
    return b"\x78\x9c" + bytes(raw) + struct.pack(">I", zlib.adler32(data) & 0xffffffff)


def chunk(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", len(payload)) + kind + payload + struct.pack(">I", zlib.crc32(kind + payload) & 0xffffffff)


def write_png(img: np.ndarray, color_type: int, filters: Union[int, Sequence[int]] = 0,
              plan: Optional[Sequence[tuple]] = None, level: int = 6, idat_sizes: Optional[Sequence[int]] = None,
              palette: Optional[np.ndarray] = None, extra_chunks: Sequence[bytes] = ()) -> bytes:
    """img [h, w, channels(color_type)] uint8 -> PNG file bytes, 8-bit, non-interlaced."""
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    assert c == CHANNELS[color_type] and img.dtype == np.uint8
    fl: List[int] = [filters] * h if isinstance(filters, int) else list(filters)
    raw = filter_rows(img, fl)
    z = deflate_segments(raw, plan or [(None, "dynamic", level)])
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
    for e in extra_chunks:
        out += e
    if color_type == 3:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).reshape(-1, 3).tobytes())
    sizes = list(idat_sizes) if idat_sizes else [len(z)]
    pos = 0
    for s in sizes:
        out += chunk(b"IDAT", z[pos:pos + s])
        pos += s
    if pos < len(z):
        out += chunk(b"IDAT", z[pos:])
    return out + chunk(b"IEND", b"")


def synth_image(rng: np.random.Generator, h: int, w: int, c: int, kind: str = "mixed") -> np.ndarray:
    """Deterministic test content: smooth ramps (good for Sub/Up/Paeth), noise (literals, long codes), flat and
    tiled regions (long and far matches)."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, c), np.int32)
    for k in range(c):
        img[:, :, k] = (xx * (3 + k) + yy * (5 - k) + 17 * k) // 2
    if kind in ("mixed", "noise"):
        m = rng.random((h, w)) < (0.12 if kind == "mixed" else 1.0)
        img[m] += rng.integers(0, 256, (int(m.sum()), c))
    if kind == "mixed":
        img[h // 3:h // 2, w // 4:3 * w // 4] = 0  # flat background, as around a hair crop
        th, tw = max(h // 8, 1), max(w // 8, 1)
        tile = rng.integers(0, 256, (th, tw, c))
        img[:th, :tw] = tile
        img[h - th:, w - tw:] = tile[:min(th, h), :min(tw, w)]
    return (img & 255).astype(np.uint8)
