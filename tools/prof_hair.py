#!/usr/bin/env python3
"""Where a HairEncoder.device_windows batch spends its time (host and device), phase by phase with a sync behind
each: python3 tools/prof_hair.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")]
import numpy as np
import torch
from bench import png_hair_files
from hcir import png, resize
from hcir.hair_encoder import HairEncoder, _sniff
from hcir.transform import knn_transform_u8

b = int(sys.argv[1]) if len(sys.argv) > 1 else 880
files = png_hair_files(12)
batch = [np.frombuffer(files[i % len(files)], np.uint8) for i in range(b)]
enc = HairEncoder(None, "vit_base_patch16", device="cuda")
blob = None
for rep in range(3):
    T = {}
    def lap(name, t0):
        torch.cuda.synchronize()
        T[name] = (time.perf_counter() - t0) * 1e3
    t = time.perf_counter(); kinds = [_sniff(a) for a in batch]; lap("sniff", t)
    t = time.perf_counter(); st = png.stage_batch(batch, threads=16, out=blob); lap("stage (16 threads, CRC)", t)
    if blob is None:
        blob = torch.empty(st.blob.numel() + 65536, dtype=torch.uint8, pin_memory=True)
    t = time.perf_counter(); d = st.to("cuda"); lap("H2D copy", t)
    t = time.perf_counter(); whole = png.decode_windows(d, (1024, 1024), _skip_rejected_check=True); lap("device decode, whole images", t)
    t = time.perf_counter(); imgs = [whole[k] for k in range(b)]; lap("python: list of views", t)
    t = time.perf_counter(); win = resize.resize_center_crop(imgs, 224); lap("resize + crop (incl. its host side)", t)
    t = time.perf_counter(); x = knn_transform_u8(win); lap("normalise", t)
    t = time.perf_counter()
    with torch.no_grad():
        f = enc.extract_features(x)
    lap("ViT-B/16 CLS", t)
    t = time.perf_counter()
    with torch.no_grad():
        f = enc.extract_features(knn_transform_u8(enc.device_windows(batch)))
    lap("device_windows + embed, as the encoder runs it", t)
    print(f"--- rep {rep}, batch {b}: " + "; ".join(f"{k} {v:.1f} ms" for k, v in T.items()), flush=True)
