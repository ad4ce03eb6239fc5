"""Generates tests/golden/png_streams.npz (run in the BUILD container; the reference cannot travel).

Contents — data only:
  * the four `assets/hair_region_only/*.png` files' bytes (the reference's own sample hair-region crops) and the
    CenterCrop(224) windows of `PIL.Image.open(...).convert("RGB")` (what HP/utils/dataloader.py:28-31 +
    HP/utils/transform.py:11 hand to the model), plus a 320-row strip of the full decode for the whole-image path;
  * seeded synthetic files from tests/png_writer.py that force every scanline filter, every deflate block type
    (stored, fixed, dynamic, mixed with matches across the seams, Huffman-only, RLE), every colour type of the
    device subset, multi-IDAT chunking, far (~30 KB) matches, odd sizes, images smaller than the window — each with
    the window Pillow decodes from it.
Usage:  python tests/golden/make_golden_png.py
"""
import glob
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from png_writer import synth_image, write_png  # noqa: E402

ASSETS = "/root/reference/assets/hair_region_only"


def pil_rgb(data: bytes) -> np.ndarray:
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def window(rgb, wh, ww):
    h, w = rgb.shape[:2]
    ph, pw = max(wh - h, 0), max(ww - w, 0)
    if ph or pw:
        rgb = np.pad(rgb, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
        h, w = rgb.shape[:2]
    top, left = int(round((h - wh) / 2.0)), int(round((w - ww) / 2.0))
    return rgb[top:top + wh, left:left + ww]


def synthetic():
    rng = np.random.default_rng(20260)
    cases = []
    for ft in range(5):  # one forced filter type each, odd size
        cases.append((f"filter{ft}_rgb", write_png(synth_image(rng, 261, 297, 3), 2, ft, level=6)))
    img = synth_image(rng, 300, 350, 4)
    cases.append(("mixed_filters_rgba_l9", write_png(img, 6, rng.integers(0, 5, 300).tolist(), level=9)))
    cases.append(("grey_l1", write_png(synth_image(rng, 280, 300, 1), 0, rng.integers(0, 5, 280).tolist(), level=1)))
    cases.append(("grey_alpha", write_png(synth_image(rng, 240, 230, 2), 4, rng.integers(0, 5, 240).tolist())))
    pal = rng.integers(0, 256, (256, 3)).astype(np.uint8)
    cases.append(("palette", write_png(synth_image(rng, 250, 260, 1), 3, rng.integers(0, 5, 250).tolist(), palette=pal)))
    cases.append(("palette_short", write_png(synth_image(rng, 230, 226, 1) % 7, 3, 1, palette=pal[:7])))
    img = synth_image(rng, 150, 160, 3, "noise")
    cases.append(("stored_blocks", write_png(img, 2, 0, plan=[(None, "stored", 0)])))
    cases.append(("fixed_blocks", write_png(synth_image(rng, 260, 250, 3), 2, 4, plan=[(None, "fixed", 6)])))
    img = synth_image(rng, 300, 320, 3)
    cases.append(("mixed_blocks", write_png(img, 2, rng.integers(0, 5, 300).tolist(), plan=[
        (40000, "dynamic", 6), (30000, "stored", 0), (50000, "fixed", 9), (1000, "stored", 0), (60000, "dynamic", 1),
        (20000, "rle", 6), (20000, "huffman", 6), (None, "dynamic", 9)])))
    cases.append(("huffman_only", write_png(synth_image(rng, 235, 240, 3), 2, 2, plan=[(None, "huffman", 6)])))
    cases.append(("rle", write_png(synth_image(rng, 235, 240, 3), 2, 1, plan=[(None, "rle", 6)])))
    # rows that repeat every 4 scanlines of 7501 bytes: matches ~30 KB back
    base = synth_image(rng, 4, 2500, 3, "noise")
    far = np.concatenate([base] * 60, 0)
    far[::7, ::11] ^= 0x55
    cases.append(("far_matches", write_png(far, 2, 0, level=9)))
    img = synth_image(rng, 700, 300, 3)
    cases.append(("tall_paeth_multi_idat", write_png(img, 2, 4, level=6, idat_sizes=[1, 0, 8192, 100, 8192, 8192])))
    cases.append(("small_100x80", write_png(synth_image(rng, 80, 100, 3), 2, rng.integers(0, 5, 80).tolist())))
    cases.append(("exact_224", write_png(synth_image(rng, 224, 224, 3), 2, 3)))
    cases.append(("one_pixel", write_png(np.array([[[9, 200, 31]]], np.uint8), 2, 0)))
    cases.append(("tall_narrow_3x900", write_png(synth_image(rng, 900, 3, 3), 2, rng.integers(0, 5, 900).tolist())))
    cases.append(("wide_short_1500x5", write_png(synth_image(rng, 5, 1500, 4), 6, rng.integers(0, 5, 5).tolist())))
    return cases


def main():
    names, blobs, wins = [], [], []
    strips = []
    for f in sorted(glob.glob(os.path.join(ASSETS, "*.png"))):
        data = open(f, "rb").read()
        rgb = pil_rgb(data)
        names.append("asset_" + os.path.basename(f))
        blobs.append(data)
        wins.append(window(rgb, 224, 224))
        strips.append(rgb[:320:5, ::3].copy())  # a thinned strip of the whole decode
    for name, data in synthetic():
        names.append(name)
        blobs.append(data)
        wins.append(window(pil_rgb(data), 224, 224))
    offs = np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64)
    out = os.path.join(HERE, "png_streams.npz")
    np.savez_compressed(out, names=np.array(names), data=np.frombuffer(b"".join(blobs), np.uint8), offsets=offs,
                        windows=np.stack(wins), asset_strips=np.stack(strips))
    print(out, len(names), "files", offs[-1], "bytes", os.path.getsize(out), "on disk")


if __name__ == "__main__":
    main()
