#!/usr/bin/env python3
"""Timing of the device PNG decoder alone: a batch of 1024x1024 RGB hair-region-like PNGs (the reference's four
sample crops from tests/golden/png_streams.npz + Pillow-written synthetic crops: textured blob on black) ->
CenterCrop(224) windows; host staging rate; host PIL rate on the same files."""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import numpy as np
import torch
from PIL import Image

from hcir import png


from bench import png_hair_files as hair_like_files  # the bench's own file set


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 880
    files = hair_like_files()
    fl = [files[i % len(files)] for i in range(batch)]
    print("file bytes/img", sum(map(len, fl)) / batch, "min/max", min(map(len, files)), max(map(len, files)))
    for th in (1, 4, 16):
        for crc in (True, False):
            t = time.perf_counter()
            st = png.stage_batch(fl, threads=th, verify_crc=crc)
            dt = time.perf_counter() - t
            print(f"stage threads={th} crc={crc}: {dt * 1e3:.1f} ms for {batch} files = {batch / dt:.0f} files/s")
    t = time.perf_counter()
    for f in files[:8]:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    print(f"host PIL decode: {(time.perf_counter() - t) / 8 * 1e3:.1f} ms per file on one core")
    d = st.to("cuda")
    for size in (224, 1024):
        if size == 1024 and batch > 64:
            d2 = png.stage_batch(fl[:64]).to("cuda")
            nb = 64
        else:
            d2, nb = d, batch
        out = png.decode_windows(d2, size, check_status=True)
        ref = np.asarray(Image.open(io.BytesIO(fl[5 % nb])).convert("RGB"))
        o = (1024 - size) // 2
        print("exact", np.array_equal(out[5 % nb].cpu().numpy(), ref[o:o + size, o:o + size]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            png.decode_windows(d2, size)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"window {size}: {ms:.3f} ms per {nb} images = {nb / ms * 1e3:.0f} img/s, "
              f"{d2.stream_bytes() / ms / 1e6:.2f} GB/s of zlib stream", flush=True)
    for nb in (1, 16, 64, 256):
        d2 = png.stage_batch(fl[:nb]).to("cuda")
        png.decode_windows(d2, 224)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            png.decode_windows(d2, 224)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"batch {nb}: {ms:.3f} ms = {nb / ms * 1e3:.0f} img/s", flush=True)


if __name__ == "__main__":
    main()
