#!/usr/bin/env python3
"""Generates tests/golden/*.npz in the BUILD container (never runs on the GPU box).

What executes here
  * the reference's own in-tree NT-Xent class, imported from
    /root/reference/experiments/DualViewHair/src/losses/ntxent_loss.py (torch only);
  * scikit-learn 1.7.2 (the library that holds the reference's kNN arithmetic):
    KNeighborsClassifier(metric="cosine").kneighbors/.predict and
    cosine_similarity + argsort[::-1] exactly as the reference calls them
    (HP/src/classification_engine.py:80-82, src/models/hair_encoder.py:193-194);
  * torch.nn.functional.normalize (HP/src/classification_engine.py:50);
  * PIL decode of the four sample JPEGs in /root/reference/assets/samples/dummy (inputs of
    config C1); the centre 224x224 windows are stored as uint8 DATA.
Only inputs and expected outputs are written; no reference source text is copied.
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_ref_ntxent():
    path = os.path.join(REF, "experiments/DualViewHair/src/losses/ntxent_loss.py")
    spec = importlib.util.spec_from_file_location("ref_ntxent_loss", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.NTXentLoss


def gen_ntxent():
    NTXentLoss = load_ref_ntxent()
    out = {}
    cases = [(8, 16, 0.5, 0), (8, 128, 0.07, 1), (64, 128, 0.7, 2), (96, 256, 0.5, 3), (50, 72, 0.2, 4)]
    for i, (b, d, t, seed) in enumerate(cases):
        g = torch.Generator().manual_seed(seed)
        z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
        loss = NTXentLoss(temperature=t)(z0, z1)
        out[f"z0_{i}"], out[f"z1_{i}"] = z0.numpy(), z1.numpy()
        out[f"t_{i}"], out[f"loss_{i}"] = np.float32(t), loss.numpy()
    # the survey's anchor value: seed-0 8x16, T=0.5 -> 2.593191623687744
    assert abs(float(out["loss_0"]) - 2.593191623687744) < 1e-6, float(out["loss_0"])
    out["n"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "ntxent_ref.npz"), **out)


def gen_knn():
    from sklearn.metrics.pairwise import cosine_similarity
    from sklearn.neighbors import KNeighborsClassifier
    out = {}
    cases = [(16, 300, 64, 5, 10), (32, 500, 96, 10, 11), (7, 50, 8, 50, 12)]
    for i, (nq, ng, d, k, seed) in enumerate(cases):
        rng = np.random.default_rng(seed)
        g = F.normalize(torch.from_numpy(rng.standard_normal((ng, d), dtype=np.float32)), dim=1)
        q = F.normalize(torch.from_numpy(rng.standard_normal((nq, d), dtype=np.float32)), dim=1)
        labels = torch.from_numpy(rng.integers(0, 7, ng))
        knn = KNeighborsClassifier(n_neighbors=k, metric="cosine")
        knn.fit(g, labels)                       # torch CPU tensors, as the reference passes them
        dist, idx = knn.kneighbors(q)
        out[f"q_{i}"], out[f"g_{i}"], out[f"labels_{i}"] = q.numpy(), g.numpy(), labels.numpy()
        out[f"k_{i}"], out[f"dist_{i}"], out[f"idx_{i}"] = np.int64(k), dist, idx
        out[f"pred_{i}"] = knn.predict(q)
        # retrieval path on UN-normalised embeddings
        gscale = rng.uniform(0.5, 3.0, (ng, 1)).astype(np.float32)
        gu = g.numpy() * gscale                  # consumers rebuild gu = g * ret_gscale
        qu = q.numpy()[0] * 2.5
        sims = cosine_similarity([qu], gu)[0]
        top = np.argsort(sims)[::-1][:k]
        out[f"ret_gscale_{i}"], out[f"ret_q_{i}"] = gscale, qu
        out[f"ret_idx_{i}"], out[f"ret_sim_{i}"] = top, sims[top]
    out["n"] = np.int64(len(cases))
    # planted exact ties: documents what the two reference paths do (NOT asserted as our tie-break)
    g = np.zeros((6, 4), dtype=np.float32)
    g[:, 0] = 1.0
    g[3:, 1] = 1.0
    qt = np.array([[1.0, 0.0, 0.0, 0.0]], dtype=np.float32)
    knn = KNeighborsClassifier(n_neighbors=3, metric="cosine").fit(g, np.arange(6))
    out["tie_g"], out["tie_q"] = g, qt
    out["tie_sklearn_idx"] = knn.kneighbors(qt)[1]
    out["tie_argsort_idx"] = np.argsort(cosine_similarity(qt, g)[0])[::-1][:3]
    np.savez_compressed(os.path.join(HERE, "knn_sklearn.npz"), **out)


def gen_normalize():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(17, 768, generator=g) * 3
    x[7] = 0.0  # eps path
    np.savez_compressed(os.path.join(HERE, "normalize.npz"), x=x.numpy(), y=F.normalize(x, dim=1).numpy())


def gen_assets():
    from PIL import Image
    d = os.path.join(REF, "assets/samples/dummy")
    wins = []
    names = sorted(os.listdir(d))
    for n in names:
        img = np.asarray(Image.open(os.path.join(d, n)).convert("RGB"))
        h, w = img.shape[:2]
        top, left = int(round((h - 224) / 2.0)), int(round((w - 224) / 2.0))
        wins.append(img[top:top + 224, left:left + 224])
    np.savez_compressed(os.path.join(HERE, "asset_windows.npz"), windows=np.stack(wins),
                        names=np.array(names), full_shape=np.array(img.shape))


if __name__ == "__main__":
    gen_ntxent()
    gen_knn()
    gen_normalize()
    gen_assets()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
