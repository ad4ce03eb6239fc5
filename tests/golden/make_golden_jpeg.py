#!/usr/bin/env python3
"""Generates tests/golden/jpeg_streams.npz in the BUILD container (never runs on the GPU box).

What executes here: Pillow 12.2 (libjpeg-turbo 3.1.4.1) — the decoder library behind both of the reference's
decode sites (torchvision.io.decode_image at HP/utils/dataloader.py:28-31, PIL at
src/models/hair_encoder.py:108).  Inputs:
  * the four sample JPEGs of /root/reference/assets/samples/dummy (config C1's inputs; 1024 x 1024 baseline
    4:2:0, no restart markers) — stored as their compressed bytes (image DATA, not source text);
  * seeded synthetic images encoded by Pillow at quality 75 / 90 / 95, subsampling 4:4:4 / 4:2:2 / 4:2:0, with
    and without restart intervals, odd sizes, one greyscale, one smaller than the window.
Expected output per stream: the CenterCrop(224) window of `Image.open(...).convert("RGB")` (zero padded when the
image is smaller, as torchvision's CenterCrop pads), uint8 [224, 224, 3].
Only inputs and expected outputs are written.
"""
import glob
import io
import os

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SIZE = 224


def window(rgb: np.ndarray, size: int = SIZE) -> np.ndarray:
    h, w = rgb.shape[:2]
    ph, pw = max(size - h, 0), max(size - w, 0)
    if ph or pw:
        rgb = np.pad(rgb, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = rgb.shape[:2]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return np.ascontiguousarray(rgb[top:top + size, left:left + size])


def synth(rng, h, w):
    """Smooth colour field + texture + a hard-edged patch: exercises long EOB runs and dense blocks."""
    base = rng.integers(0, 256, (h // 16 + 2, w // 16 + 2, 3)).astype(np.uint8)
    a = np.asarray(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int16)
    a[: h // 2] += rng.integers(-24, 24, (h // 2, w, 3), dtype=np.int16)
    a[h // 3: h // 2, w // 4: w // 2] = rng.integers(0, 256, 3)
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


def main():
    rng = np.random.default_rng(20260104)
    names, blobs = [], []
    for f in sorted(glob.glob(os.path.join(REF, "assets/samples/dummy/*.jpg"))):
        names.append("asset/" + os.path.basename(f))
        blobs.append(open(f, "rb").read())
    cases = [  # (h, w, quality, subsampling, restart_marker_blocks, grey)
        (384, 512, 75, 2, 0, False), (384, 512, 90, 2, 0, False), (384, 512, 95, 2, 0, False),
        (300, 420, 90, 1, 0, False), (300, 420, 90, 0, 0, False),
        (333, 257, 75, 2, 4, False), (333, 257, 95, 1, 1, False), (260, 280, 90, 0, 7, False),
        (227, 229, 95, 2, 0, False), (640, 480, 90, 2, 2, False),
        (300, 300, 90, 0, 0, True), (120, 500, 90, 2, 0, False), (96, 80, 75, 2, 3, False),
    ]
    for h, w, q, ss, ri, grey in cases:
        im = synth(rng, h, w)
        kw = dict(quality=q)
        if grey:
            im = im.convert("L")
        else:
            kw["subsampling"] = ss
        if ri:
            kw["restart_marker_blocks"] = ri
        b = io.BytesIO()
        im.save(b, "JPEG", **kw)
        names.append(f"synth/{h}x{w}_q{q}_ss{ss}_ri{ri}{'_grey' if grey else ''}")
        blobs.append(b.getvalue())
    # one progressive and one PNG: the stager must reject them (HCIR_ERR_UNSUPPORTED / INVALID)
    b = io.BytesIO()
    synth(rng, 240, 256).save(b, "JPEG", quality=85, progressive=True)
    names.append("reject/progressive")
    blobs.append(b.getvalue())
    b = io.BytesIO()
    synth(rng, 64, 64).save(b, "PNG")
    names.append("reject/png")
    blobs.append(b.getvalue())
    wins = np.stack([window(np.asarray(Image.open(io.BytesIO(x)).convert("RGB"))) for x in blobs])
    offs = np.cumsum([0] + [len(x) for x in blobs]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "jpeg_streams.npz"),
                        names=np.array(names), data=np.frombuffer(b"".join(blobs), dtype=np.uint8), offsets=offs,
                        windows=wins, decoder=np.array(f"Pillow libjpeg-turbo {features.version('libjpeg_turbo')}"))
    print(len(names), "streams,", offs[-1], "compressed bytes ->", os.path.getsize(os.path.join(HERE, "jpeg_streams.npz")))


if __name__ == "__main__":
    main()
