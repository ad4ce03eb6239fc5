#!/usr/bin/env python3
"""Timing of the device JPEG decoder alone: batch of synthetic 1024x1024 baseline 4:2:0 files -> CenterCrop(224)
windows; host staging rate; per-kernel split by HIP events is left to rocprofv3 (tools/profile_round.sh)."""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import numpy as np
import torch
from PIL import Image

from hcir import jpeg


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 880
    rng = np.random.default_rng(7)
    files = []
    for _ in range(32):
        base = rng.integers(0, 256, (40, 40, 3)).astype(np.uint8)
        a = np.asarray(Image.fromarray(base).resize((1024, 1024), Image.BICUBIC)).astype(np.int16)
        a[:, :512] += rng.integers(-12, 12, (1024, 512, 3), dtype=np.int16)
        b = io.BytesIO()
        Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=88, subsampling=2)
        files.append(b.getvalue())
    fl = [files[i % 32] for i in range(batch)]
    print("file bytes/img", sum(map(len, fl)) / batch)
    for th in (1, 4, 16):
        t = time.perf_counter()
        st = jpeg.stage_batch(fl, threads=th)
        print(f"stage threads={th}: {(time.perf_counter() - t) * 1e3:.1f} ms for {batch} files")
    d = st.to("cuda")
    for size in (224, 1024):
        if size == 1024 and batch > 64:
            d2 = jpeg.stage_batch(fl[:64]).to("cuda")
            nb = 64
        else:
            d2, nb = d, batch
        out = jpeg.decode_windows(d2, size, check_status=True)
        ref = np.asarray(Image.open(io.BytesIO(fl[3])).convert("RGB"))
        o = (1024 - size) // 2
        print("exact", np.array_equal(out[3].cpu().numpy(), ref[o:o + size, o:o + size]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            jpeg.decode_windows(d2, size)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"window {size}: {ms:.3f} ms per {nb} images = {nb / ms * 1e3:.0f} img/s, "
              f"{d2.stream_bytes() / ms / 1e6:.2f} GB/s of entropy-coded stream")


if __name__ == "__main__":
    main()
