"""hcir.index — the FAISS calls of the reference's other retrieval front-end on the MI355X scan
(SURVEY.md §8f rank 1).

The gradio app and the DualViewHair evaluation build `faiss.IndexFlatL2` over L2-normalised
embeddings and call `.search(feat, k)` (HP/app/inference.py:75,90-96,108;
experiments/DualViewHair/scripts/quantitative_eval.py:116,143-150,183-186).  faiss is not a dependency;
the same interface is provided on top of hcir_sim_topk:

    normalize_L2(x)                 in place, like faiss.normalize_L2
    IndexFlatL2(d).add(x)           rows kept resident in HBM
    D, I = index.search(x, k)       squared L2 distances ascending, int64 ids, -1 / +inf padding
    write_index / read_index        embeddings + ntotal in one .npz (NOT faiss's binary format)

Squared L2 is ranked exactly, for any (not only unit-norm) vectors, by the inner product of
augmented rows  g' = [g, ||g||^2, 0..],  q' = [2q, -1, 0..]  (8 extra columns keep d % 8 == 0):
q'.g' = 2 q.g - ||g||^2, so  D = ||q||^2 - q'.g'.  Ties: smaller id first.
"""
from __future__ import annotations

import pickle
from typing import List, Tuple

import numpy as np
import torch

from . import ops
from ._lib import HcirError


def normalize_L2(x: np.ndarray) -> None:
    """faiss.normalize_L2: in-place row normalisation of a float32 matrix (zero rows stay zero)."""
    if x.dtype != np.float32 or x.ndim != 2:
        raise TypeError("normalize_L2 expects a 2-D float32 array")
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    np.divide(x, n, out=x, where=n > 0)


class IndexFlatL2:
    def __init__(self, d: int, device="cuda"):
        if d <= 0:
            raise ValueError("dimension must be positive")
        self.d = int(d)
        self.device = torch.device(device)
        self.is_trained = True
        self._chunks: List[torch.Tensor] = []
        self._aug = None

    @property
    def ntotal(self) -> int:
        return sum(c.shape[0] for c in self._chunks)

    def reset(self) -> None:
        self._chunks, self._aug = [], None

    def add(self, x) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] float32, got {x.shape}")
        self._chunks.append(torch.from_numpy(x).to(self.device))
        self._aug = None

    def _augmented(self) -> torch.Tensor:
        if self._aug is None:
            g = torch.cat(self._chunks, 0)
            dp = (self.d + 7) // 8 * 8 + 8
            aug = torch.zeros((g.shape[0], dp), dtype=torch.float32, device=self.device)
            aug[:, : self.d] = g
            aug[:, dp - 8] = (g.double() ** 2).sum(1).float()
            self._aug, self._chunks = aug, [g]
        return self._aug

    def reconstruct_n(self, i0: int = 0, n: int = -1) -> np.ndarray:
        g = torch.cat(self._chunks, 0)
        return g[i0: (None if n < 0 else i0 + n)].cpu().numpy()

    def search(self, x, k: int) -> Tuple[np.ndarray, np.ndarray]:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] float32, got {x.shape}")
        nq = x.shape[0]
        D = np.full((nq, k), np.inf, dtype=np.float32)   # faiss pads missing results with +inf / -1
        I = np.full((nq, k), -1, dtype=np.int64)
        if self.ntotal == 0 or nq == 0:
            return D, I
        aug = self._augmented()
        dp = aug.shape[1]
        q = torch.zeros((nq, dp), dtype=torch.float32, device=self.device)
        xt = torch.from_numpy(x).to(self.device)
        q[:, : self.d] = 2.0 * xt
        q[:, dp - 8] = -1.0
        kk = min(k, self.ntotal)
        if kk > 1024:
            raise ValueError("k > 1024 (HCIR_TOPK_MAX) is not supported by hcir_sim_topk")
        val, idx = ops.sim_topk(q, aug, kk)
        qn = (xt.double() ** 2).sum(1, keepdim=True)
        D[:, :kk] = (qn - val.double()).clamp_(min=0.0).float().cpu().numpy()
        I[:, :kk] = idx.cpu().numpy()
        return D, I


def write_index(index: IndexFlatL2, path: str) -> None:
    np.savez(path if path.endswith(".npz") else path + ".npz", d=index.d, x=index.reconstruct_n())


def read_index(path: str, device="cuda") -> IndexFlatL2:
    z = np.load(path if path.endswith(".npz") else path + ".npz")
    index = IndexFlatL2(int(z["d"]), device=device)
    if len(z["x"]):
        index.add(z["x"])
    return index


def save_paths(paths: List[str], path: str) -> None:
    """The reference pickles its path list next to the index (HP/app/inference.py:97-98)."""
    with open(path, "wb") as f:
        pickle.dump(list(paths), f)


def load_paths(path: str) -> List[str]:
    with open(path, "rb") as f:
        return pickle.load(f)
