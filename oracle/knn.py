"""oracle.knn — CPU restatement of the reference's brute-force cosine kNN.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference call sites restated here
  * Classifier.knn_eval -> sklearn KNeighborsClassifier(metric="cosine")
        HP/src/classification_engine.py:79-82
  * HairEncoder.retrieve_similar_images -> cosine_similarity + argsort[::-1][:k]
        src/models/hair_encoder.py:193-194
  * NegSamplerStatic -> normalise, mm, sort desc, pick column k-1
        HP/src/neg_sampling.py:31-53
scikit-learn is un-pinned by the reference (requirements.txt:1-8); 1.7.2 is what
is installed and what tests/golden/make_golden.py ran.

Two layers:
  * numpy float64 functions (`*_np`) — the mathematical definition;
  * ctypes wrappers over oracle/knn_oracle.c — the bit-level restatement of
    libhcir's fp32 fmaf-chain order, and a multi-threaded CPU baseline.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_CHAIN32 = 0
MODE_F64 = 1
MODE_CHAIN32_SCALAR = 2  # dot_chain32 one score at a time (the definition the blocked AVX2 path is checked against)


def build() -> str:
    """Compile knn_oracle.c (gcc) if the .so is missing or stale; return its path."""
    so = os.path.join(_HERE, "libknn_oracle.so")
    src = os.path.join(_HERE, "knn_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.hcir_oracle_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(t))


def num_threads() -> int:
    return int(lib().hcir_oracle_num_threads())


def row_invnorm(x: np.ndarray, eps: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape[0], dtype=np.float32)
    lib().hcir_oracle_row_invnorm(
        _p(x, ctypes.c_float), ctypes.c_int64(x.shape[0]), ctypes.c_int32(x.shape[1]),
        ctypes.c_int64(x.shape[1]), ctypes.c_float(eps), _p(out, ctypes.c_float))
    return out


def cosine_topk(q, g, k, qn=None, gn=None, idx_base=0, mode=MODE_CHAIN32):
    """(values fp32 [nq,k], indices int64 [nq,k]); score = (<g,q> * gn) * qn."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    g = np.ascontiguousarray(g, dtype=np.float32)
    assert q.shape[1] == g.shape[1]
    qn = None if qn is None else np.ascontiguousarray(qn, dtype=np.float32)
    gn = None if gn is None else np.ascontiguousarray(gn, dtype=np.float32)
    val = np.empty((q.shape[0], k), dtype=np.float32)
    idx = np.empty((q.shape[0], k), dtype=np.int64)
    lib().hcir_oracle_cosine_topk(
        _p(q, ctypes.c_float), ctypes.c_int64(q.shape[0]), _p(g, ctypes.c_float),
        ctypes.c_int64(g.shape[0]), ctypes.c_int32(q.shape[1]), ctypes.c_int32(k),
        _p(qn, ctypes.c_float), _p(gn, ctypes.c_float), ctypes.c_int64(idx_base),
        ctypes.c_int(mode), _p(val, ctypes.c_float), _p(idx, ctypes.c_int64))
    return val, idx


def scores(q, g, qn=None, gn=None, mode=MODE_CHAIN32):
    q = np.ascontiguousarray(q, dtype=np.float32)
    g = np.ascontiguousarray(g, dtype=np.float32)
    qn = None if qn is None else np.ascontiguousarray(qn, dtype=np.float32)
    gn = None if gn is None else np.ascontiguousarray(gn, dtype=np.float32)
    out = np.empty((q.shape[0], g.shape[0]), dtype=np.float32)
    lib().hcir_oracle_scores(
        _p(q, ctypes.c_float), ctypes.c_int64(q.shape[0]), _p(g, ctypes.c_float),
        ctypes.c_int64(g.shape[0]), ctypes.c_int32(q.shape[1]), _p(qn, ctypes.c_float),
        _p(gn, ctypes.c_float), ctypes.c_int(mode), _p(out, ctypes.c_float))
    return out


def topk_merge(vals, idx, k_out):
    """vals/idx: [nlists, nq, k_in] sorted lists -> ([nq,k_out], [nq,k_out])."""
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    nl, nq, kin = vals.shape
    ov = np.empty((nq, k_out), dtype=np.float32)
    oi = np.empty((nq, k_out), dtype=np.int64)
    lib().hcir_oracle_topk_merge(
        _p(vals, ctypes.c_float), _p(idx, ctypes.c_int64), ctypes.c_int32(nl),
        ctypes.c_int64(nq), ctypes.c_int32(kin), ctypes.c_int32(k_out),
        _p(ov, ctypes.c_float), _p(oi, ctypes.c_int64))
    return ov, oi


def knn_vote(nbr_idx, labels, nclass=None):
    """sklearn uniform-weight vote: mode of the k neighbour labels, smallest label on ties."""
    nbr_idx = np.ascontiguousarray(nbr_idx, dtype=np.int64)
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    nclass = int(labels.max()) + 1 if nclass is None else int(nclass)
    pred = np.empty(nbr_idx.shape[0], dtype=np.int64)
    lib().hcir_oracle_knn_vote(
        _p(nbr_idx, ctypes.c_int64), ctypes.c_int64(nbr_idx.shape[0]),
        ctypes.c_int32(nbr_idx.shape[1]), _p(labels, ctypes.c_int64), ctypes.c_int64(nclass),
        _p(pred, ctypes.c_int64))
    return pred


# ---------------------------------------------------------------------------
# numpy float64 definitions (no C): the mathematical statement of each call site
# ---------------------------------------------------------------------------
def stable_topk_np(scores64: np.ndarray, k: int):
    """Top-k of each row under (score desc, index asc)."""
    n = scores64.shape[1]
    if n <= 4096 or k * 8 >= n:
        order = np.argsort(-scores64, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(scores64, order, axis=1), order.astype(np.int64)
    # wide rows: every element >= the k-th largest value (ties at the boundary included), then the
    # same stable order on that subset — identical result, no full-row sort
    vals = np.empty((scores64.shape[0], k), dtype=scores64.dtype)
    idx = np.empty((scores64.shape[0], k), dtype=np.int64)
    for i, row in enumerate(scores64):
        kth = np.partition(row, n - k)[n - k]
        cand = np.nonzero(row >= kth)[0]                 # ascending indices
        o = cand[np.argsort(-row[cand], kind="stable")[:k]]
        vals[i], idx[i] = row[o], o
    return vals, idx


def sklearn_cosine_kneighbors_np(q, g, k):
    """KNeighborsClassifier(metric='cosine').kneighbors restated:
    S = normalize(X) normalize(Y)^T; d = clip(1 - S, 0, 2); k smallest
    (sklearn/metrics/pairwise.py cosine_distances; neighbors/_base.py:722-761)."""
    q = np.asarray(q, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    qh = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-300)
    gh = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-300)
    s = qh @ gh.T
    val, idx = stable_topk_np(s, k)
    return np.clip(1.0 - val, 0.0, 2.0), idx


def retrieve_similar_np(query, gallery, k):
    """HairEncoder.retrieve_similar_images (src/models/hair_encoder.py:193-194):
    cosine_similarity([q], G)[0] then the k largest.  The reference's
    argsort()[::-1] puts the HIGHEST index first among exact ties; this build fixes
    (score desc, index asc) instead."""
    d, idx = sklearn_cosine_kneighbors_np(np.asarray(query)[None, :], gallery, k)
    return 1.0 - d[0], idx[0]


def neg_sampler_static_np(emb, k):
    """NegSamplerStatic (HP/src/neg_sampling.py:31-53): column k-1 of the row-wise
    descending sort of the B x B cosine matrix (self is rank 0)."""
    emb = np.asarray(emb, dtype=np.float64)
    b = emb.shape[0]
    if k < 1 or k > b:
        raise ValueError(f"k must be between 1 and {b}")
    en = emb / np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-8)
    s = en @ en.T
    _, idx = stable_topk_np(s, k)
    return idx[:, k - 1]
