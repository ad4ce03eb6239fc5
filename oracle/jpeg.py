"""oracle.jpeg — TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of the baseline-JPEG decode that the reference's loader runs before `knn_transform`:
`HairPretraining/utils/dataloader.py:28-31` (`read_file` -> `torchvision.io.decode_image(..., RGB)`) and
`src/models/hair_encoder.py:108,169` (`PIL.Image.open(...).convert('RGB')`).  Both hand the bytes to
libjpeg(-turbo) with its default decompression parameters: `dct_method = JDCT_ISLOW`,
`do_fancy_upsampling = TRUE`, output colour space RGB.  libjpeg-turbo is a third-party dependency that is
NOT under /root/reference; the arithmetic restated here is its published algorithm (file names of
libjpeg-turbo 3.1):
  * jdhuff.c      sequential Huffman decode (DC differences, AC run/size, EOB, ZRL, restart intervals)
  * jidctint.c    `jpeg_idct_islow`: CONST_BITS 13, PASS1_BITS 2, range limit through `& RANGE_MASK`
  * jdsample.c    `h2v1_fancy_upsample`, `h2v2_fancy_upsample` (triangle filter, alternating rounding),
                  edge replication as set up by jdmainct.c (`set_bottom_pointers`, first-row context)
  * jdcolor.c     `ycc_rgb_convert` with the 16-bit fixed-point tables of `build_ycc_rgb_table`
PIN: Pillow 12.2 in this image bundles libjpeg-turbo 3.1.4.1; `tests/test_jpeg_oracle.py` checks this
restatement byte for byte against `PIL.Image.open(...).convert("RGB")` on the four asset JPEGs' goldens and on
seeded synthetic streams (4:4:4 / 4:2:2 / 4:2:0 / grey, restart intervals, odd sizes).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s CPU-baseline leg may import this module.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
    54, 47, 55, 62, 63], dtype=np.int64)


class Unsupported(ValueError):
    """Stream is outside the baseline subset (progressive, arithmetic, 12-bit, CMYK, odd sampling)."""


def parse(data: bytes) -> Dict:
    """Marker walk (jdmarker.c): tables, frame, scan header, entropy-coded segment boundaries."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG (no SOI)")
    i = 2
    qt: Dict[int, np.ndarray] = {}
    huff: Dict[Tuple[int, int], Tuple[List[int], List[int]]] = {}
    info: Dict = {"restart_interval": 0}
    while True:
        if data[i] != 0xFF:
            raise Unsupported(f"marker expected at {i}")
        while data[i + 1] == 0xFF:  # fill bytes
            i += 1
        m = data[i + 1]
        i += 2
        if m == 0xD9:
            raise Unsupported("EOI before SOS")
        (ln,) = struct.unpack(">H", data[i:i + 2])
        seg = data[i + 2:i + ln]
        if m == 0xDB:
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                if pq:
                    vals = struct.unpack(">64H", seg[j + 1:j + 129])
                    j += 129
                else:
                    vals = struct.unpack("64B", seg[j + 1:j + 65])
                    j += 65
                nat = np.zeros(64, dtype=np.int64)
                nat[ZIGZAG] = vals  # file order is zigzag; keep natural (row-major) order
                qt[tq] = nat
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                bits = list(seg[j + 1:j + 17])
                n = sum(bits)
                huff[(tc, th)] = (bits, list(seg[j + 17:j + 17 + n]))
                j += 17 + n
        elif m in (0xC0, 0xC1):
            p, h, w, nc = struct.unpack(">BHHB", seg[:6])
            if p != 8:
                raise Unsupported("sample precision != 8")
            info.update(height=h, width=w, ncomp=nc,
                        comps=[dict(id=seg[6 + 3 * k], h=seg[7 + 3 * k] >> 4, v=seg[7 + 3 * k] & 15, tq=seg[8 + 3 * k])
                               for k in range(nc)])
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported(f"SOF type 0x{m:02X} (progressive / lossless / arithmetic)")
        elif m == 0xDD:
            (info["restart_interval"],) = struct.unpack(">H", seg[:2])
        elif m == 0xDA:
            ns = seg[0]
            if "comps" not in info or ns != info["ncomp"]:
                raise Unsupported("non-interleaved / multi-scan stream")
            for k in range(ns):
                cid, t = seg[1 + 2 * k], seg[2 + 2 * k]
                comp = next(c for c in info["comps"] if c["id"] == cid)
                comp["td"], comp["ta"] = t >> 4, t & 15
            i += ln
            break
        i += ln
    info["qt"], info["huff"] = qt, huff
    # entropy-coded data: up to the marker that is neither RSTn nor a stuffed zero
    segs: List[bytes] = []
    cur = bytearray()
    n = len(data)
    while i < n:
        b = data[i]
        if b != 0xFF:
            cur.append(b)
            i += 1
            continue
        nb = data[i + 1] if i + 1 < n else 0xD9
        if nb == 0x00:
            cur.append(0xFF)
            i += 2
        elif 0xD0 <= nb <= 0xD7:
            segs.append(bytes(cur))
            cur = bytearray()
            i += 2
        elif nb == 0xFF:
            i += 1
        else:
            break
    segs.append(bytes(cur))
    info["segments"] = segs
    return info


def _derive(bits: List[int], vals: List[int]):
    """jdhuff.c jpeg_make_d_derived_tbl: code -> (length, symbol) map as a dict keyed by (length, code)."""
    table = {}
    code, k = 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            table[(ln, code)] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self, data: bytes):
        self.d, self.pos, self.n = data, 0, len(data) * 8

    def bit(self) -> int:
        p = self.pos
        self.pos = p + 1
        if p >= self.n:
            return 0  # libjpeg feeds zeros past the end of a segment (with a warning)
        return (self.d[p >> 3] >> (7 - (p & 7))) & 1

    def get(self, n: int) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def _decode_sym(br: _Bits, table) -> int:
    code = 0
    for ln in range(1, 17):
        code = (code << 1) | br.bit()
        s = table.get((ln, code))
        if s is not None:
            return s
    raise ValueError("bad Huffman code")


def _extend(v: int, s: int) -> int:
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1  # HUFF_EXTEND


def decode_coefficients(info: Dict) -> List[np.ndarray]:
    """Per component: int64 [blocks_y, blocks_x, 64] quantised coefficients in NATURAL order, DC predictions
    resolved (jdhuff.c decode_mcu_slow; MCU-interleaved scan, restart intervals reset the DC predictors)."""
    comps = info["comps"]
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    mx = -(-info["width"] // (8 * hmax))
    my = -(-info["height"] // (8 * vmax))
    if info["ncomp"] == 1:  # a single-component scan is never interleaved: MCU = one block
        hmax = vmax = 1
        comps = [dict(comps[0], h=1, v=1)]
        mx, my = -(-info["width"] // 8), -(-info["height"] // 8)
    out = [np.zeros((my * c["v"], mx * c["h"], 64), dtype=np.int64) for c in comps]
    tabs = {k: _derive(*v) for k, v in info["huff"].items()}
    ri = info["restart_interval"] or mx * my
    mcu = 0
    for seg in info["segments"]:
        br = _Bits(seg)
        pred = [0] * len(comps)
        for _ in range(min(ri, mx * my - mcu)):
            yy, xx = divmod(mcu, mx)
            for ci, c in enumerate(comps):
                dct, act = tabs[(0, c["td"])], tabs[(1, c["ta"])]
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = out[ci][yy * c["v"] + by, xx * c["h"] + bx]
                        s = _decode_sym(br, dct)
                        if s:
                            pred[ci] += _extend(br.get(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = _decode_sym(br, act)
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                blk[ZIGZAG[k]] = _extend(br.get(s), s)
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
            mcu += 1
    info["_geom"] = dict(hmax=hmax, vmax=vmax, mx=mx, my=my, comps=comps)
    return out


# ---- jidctint.c : jpeg_idct_islow ---------------------------------------------------------------------------------
_CB, _P1 = 13, 2
_F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137,
          f1_961=16069, f2_053=16819, f2_562=20995, f3_072=25172)


def _idct_1d(x, shift):
    """One pass of the LL&M 8-point IDCT over the LAST axis of x (int64 [..., 8]); DESCALE by `shift`."""
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * _F["f0_541"]
    tmp2 = z1 + z3 * (-_F["f1_847"])
    tmp3 = z1 + z2 * _F["f0_765"]
    z2, z3 = x[..., 0], x[..., 4]
    tmp0 = (z2 + z3) << _CB
    tmp1 = (z2 - z3) << _CB
    tmp10, tmp13 = tmp0 + tmp3, tmp0 - tmp3
    tmp11, tmp12 = tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * _F["f1_175"]
    t0 = t0 * _F["f0_298"]
    t1 = t1 * _F["f2_053"]
    t2 = t2 * _F["f3_072"]
    t3 = t3 * _F["f1_501"]
    z1 = z1 * (-_F["f0_899"])
    z2 = z2 * (-_F["f2_562"])
    z3 = z3 * (-_F["f1_961"]) + z5
    z4 = z4 * (-_F["f0_390"]) + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    rnd = 1 << (shift - 1)
    out = np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3],
                   axis=-1)
    return (out + rnd) >> shift  # arithmetic shift: numpy int64 >> floors, as RIGHT_SHIFT does


def range_limit(v):
    """`range_limit[v & RANGE_MASK]` of jidctint.c with jdmaster.c prepare_range_limit_table's post-IDCT table."""
    i = np.asarray(v) & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_islow(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """[..., 64] quantised coefficients (natural order) * q[64] -> uint8 samples [..., 8, 8]."""
    x = (coef * q).reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(x, -1, -2), _CB - _P1)      # pass 1: columns (transform along the row index)
    ws = np.swapaxes(ws, -1, -2)
    px = _idct_1d(ws, _CB + _P1 + 3)                        # pass 2: rows
    return range_limit(px)


# ---- jdsample.c ----------------------------------------------------------------------------------------------------
def _plane(samples: np.ndarray) -> np.ndarray:
    by, bx = samples.shape[:2]
    return samples.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def upsample_h2v1(p: np.ndarray) -> np.ndarray:
    """h2v1_fancy_upsample on rows of the REAL downsampled width (callers slice first)."""
    p = p.astype(np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out = np.empty((p.shape[0], p.shape[1] * 2), dtype=np.int64)
    out[:, 0::2] = (p * 3 + left + 1) >> 2
    out[:, 1::2] = (p * 3 + right + 2) >> 2
    return out.astype(np.uint8)


def upsample_h2v2(p: np.ndarray) -> np.ndarray:
    """h2v2_fancy_upsample; rows above the first / below the last real row replicate (jdmainct.c)."""
    p = p.astype(np.int64)
    up = np.concatenate([p[:1], p[:-1]], axis=0)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((p.shape[0] * 2, p.shape[1] * 2), dtype=np.int64)
    for v, other in ((0, up), (1, dn)):
        cs = p * 3 + other                                  # thiscolsum
        left = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1)
        right = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        out[v::2, 0::2] = (cs * 3 + left + 8) >> 4
        out[v::2, 1::2] = (cs * 3 + right + 7) >> 4
    return out.astype(np.uint8)


# ---- jdcolor.c -----------------------------------------------------------------------------------------------------
def _fix(x):
    return int(x * 65536 + 0.5)


_X = np.arange(256, dtype=np.int64) - 128
CR_R = (_fix(1.40200) * _X + 32768) >> 16
CB_B = (_fix(1.77200) * _X + 32768) >> 16
CR_G = -_fix(0.71414) * _X
CB_G = -_fix(0.34414) * _X + 32768


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    y = y.astype(np.int64)
    r = np.clip(y + CR_R[cr], 0, 255)
    g = np.clip(y + ((CB_G[cb] + CR_G[cr]) >> 16), 0, 255)
    b = np.clip(y + CB_B[cb], 0, 255)
    return np.stack([r, g, b], axis=-1).astype(np.uint8)


def decode(data: bytes, return_stages: bool = False):
    """Baseline JPEG bytes -> uint8 [H, W, 3] RGB exactly as libjpeg-turbo's default decompressor emits it."""
    info = parse(data)
    coefs = decode_coefficients(info)
    g = info["_geom"]
    h, w = info["height"], info["width"]
    planes = []
    for ci, c in enumerate(g["comps"]):
        samples = idct_islow(coefs[ci], info["qt"][c["tq"]])
        full = _plane(samples)
        dw = -(-w * c["h"] // g["hmax"])
        dh = -(-h * c["v"] // g["vmax"])
        real = full[:dh, :dw]
        fx, fy = g["hmax"] // c["h"], g["vmax"] // c["v"]
        if (fx, fy) == (1, 1):
            up = real
        elif (fx, fy) in ((2, 1), (2, 2)) and dw <= 2:
            # jdsample.c jinit_upsampler: fancy upsampling needs downsampled_width > 2, else box replication
            up = np.repeat(np.repeat(real, fy, axis=0), fx, axis=1)
        elif (fx, fy) == (2, 1):
            up = upsample_h2v1(real)
        elif (fx, fy) == (2, 2):
            up = upsample_h2v2(real)
        else:
            raise Unsupported(f"sampling ratio {fx}x{fy}")
        planes.append(up[:h, :w])
    if len(planes) == 1:
        rgb = np.repeat(planes[0][..., None], 3, axis=-1)
    elif len(planes) == 3:
        rgb = ycc_to_rgb(*planes)
    else:
        raise Unsupported("4-component (CMYK / YCCK) stream")
    if return_stages:
        return rgb, dict(info=info, coefs=coefs, planes=planes)
    return rgb


def center_window(rgb: np.ndarray, size: int = 224) -> np.ndarray:
    """torchvision CenterCrop(size) on a decoded image (zero padding when smaller), HP/utils/transform.py:11."""
    h, w = rgb.shape[:2]
    ph, pw = max(size - h, 0), max(size - w, 0)
    if ph or pw:
        rgb = np.pad(rgb, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = rgb.shape[:2]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return np.ascontiguousarray(rgb[top:top + size, left:left + size])
