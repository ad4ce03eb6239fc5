#!/usr/bin/env python3
"""Where a Huffman workgroup spends its time: run against a library built with -DHCIR_JPEG_STAMPS
(tools/build_variant.sh stamps "-DHCIR_JPEG_STAMPS" jpeg.hip; HCIR_LIB_PATH=tools/_libhcir_stamps.so).
Reads the per-image geometry records (first bytes of the workspace) back: hand-over rounds and the s_memrealtime
stamps (100 MHz) at the phase boundaries."""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import numpy as np
import torch
from PIL import Image

from hcir import jpeg

GEOM_WORDS = 45  # sizeof(JGeom) / 4: JWin 8, 3 x JPlane 8, plane_off 3, ncomp/width/height 3, rounds 1, stamp 6


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
    d = jpeg.stage_batch([files[i % 32] for i in range(batch)]).to("cuda")
    for _ in range(3):
        jpeg.decode_windows(d, 224)
    torch.cuda.synchronize()
    ws = next(iter(jpeg._ws.values()))
    base = (-ws.data_ptr()) % 256
    words = None
    for gw in (GEOM_WORDS, GEOM_WORDS - 1, GEOM_WORDS + 1):
        g = ws[base:base + batch * gw * 4].cpu().numpy().view(np.int32).reshape(batch, gw)
        if (g[:, 35] == 3).all() and (g[:, 36] == 1024).all():
            words = gw
            break
    assert words, "geometry record layout not recognised"
    rounds = g[:, 38]
    st = g[:, 39:45].astype(np.int64) & 0xFFFFFFFF
    dt = np.diff(st, axis=1) / 100.0  # us
    print(f"batch {batch}: rounds min/median/max {rounds.min()} / {int(np.median(rounds))} / {rounds.max()}")
    if st.any():
        names = ["phase 0 (speculative)", "round 1 (+scan)", "rounds 2.. (compacted)", "scan + write pass", "DC scan"]
        for i, n in enumerate(names):
            print(f"  {n:26s} median {np.median(dt[:, i]):8.1f} us   max {dt[:, i].max():8.1f} us")
        print(f"  workgroup total            median {np.median(st[:, 5] - st[:, 0]) / 100:8.1f} us   max {(st[:, 5] - st[:, 0]).max() / 100:8.1f} us")
        print(f"  first start .. last end    {(st[:, 5].max() - st[:, 0].min()) / 100:8.1f} us")


if __name__ == "__main__":
    main()
