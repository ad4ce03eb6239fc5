#!/usr/bin/env python3
"""One-off wider sweep of the device PNG decoder against Pillow: random sizes, colour types, compression levels and
contents (noise, smooth, flat, sparse), whole-image windows.  python3 tools/sweep_png.py [count] [seed]"""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import numpy as np
import torch
from PIL import Image
from hcir import png

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = skipped = 0
for i in range(n):
    h, w = int(rng.integers(1, 700)), int(rng.integers(1, 900))
    mode = ["L", "LA", "RGB", "RGBA", "P"][int(rng.integers(0, 5))]
    ch = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4, "P": 1}[mode]
    kind = int(rng.integers(0, 5))
    if kind == 0:
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([((yy * (c + 1) + xx * (3 - c)) // 3) % 256 for c in range(ch)], 2).astype(np.uint8)
    elif kind == 2:
        a = np.full((h, w, ch), int(rng.integers(0, 256)), np.uint8)
    elif kind == 3:
        a = np.zeros((h, w, ch), np.uint8)
        m = rng.random((h, w)) < 0.03
        a[m] = rng.integers(0, 256, (int(m.sum()), ch), dtype=np.uint8)
    else:
        base = rng.integers(0, 250, (h // 16 + 1, w // 16 + 1, ch), dtype=np.uint8)
        a = np.kron(base, np.ones((16, 16, 1), np.uint8))[:h, :w] + rng.integers(0, 6, (h, w, ch), dtype=np.uint8)
    im = Image.fromarray(a[:, :, 0] if ch == 1 else a, "L" if mode == "P" else mode)
    if mode == "P":
        im = im.convert("P", palette=Image.ADAPTIVE, colors=int(rng.integers(2, 257)))
    buf = io.BytesIO()
    im.save(buf, "PNG", compress_level=int(rng.integers(0, 10)), optimize=bool(rng.integers(0, 2)))
    f = buf.getvalue()
    want = np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    st = png.stage_batch([f])
    if st.rejected:
        depth = f[24]                      # IHDR bit depth: Pillow packs palettes of <= 16 colours into 1 / 2 / 4 bits
        if st.status[0] == -2 and depth < 8:
            skipped += 1                   # outside the device subset (8-bit only): the loader's host path, by design
        else:
            print(i, mode, (h, w), "rejected by the stager", st.status); bad += 1
        continue
    got = png.decode_windows(st.to("cuda"), (h, w), check_status=True)[0].cpu().numpy()
    if not np.array_equal(got, want):
        print(i, mode, (h, w), "kind", kind, "MISMATCH", int((got != want).sum())); bad += 1
print(f"{n} files: {n - bad - skipped} byte-identical to Pillow, {skipped} sub-8-bit palette files left to the host, {bad} bad")
sys.exit(1 if bad else 0)
