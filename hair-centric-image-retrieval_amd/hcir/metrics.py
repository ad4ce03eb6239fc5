"""hcir.metrics — class vote and retrieval metrics on the device.

  knn_vote(nbr_idx, labels, ks, nclass)    sklearn KNeighborsClassifier.predict for every k of the sweep from
                                           one top-max(k) list (HP/src/classification_engine.py:71,79-82)
  confusion_matrix(y_true, y_pred, nclass) counts of sklearn's confusion_matrix (:85); accuracy = trace / n (:83)
  retrieval_metrics(retrieved, gt, ks)     Recall@K / mAP@K of experiments/DualViewHair/scripts/
                                           quantitative_eval.py:194-209,228-234
Inputs and outputs are HIP-device tensors; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Sequence, Tuple

import torch

from . import _lib
from ._lib import HcirError, check
from .ops import _dev, _stream


def _ks(ks: Sequence[int], kmax: int):
    ks = [int(k) for k in ks]
    if not ks or len(ks) > 16 or any(k < 1 or k > kmax for k in ks) or any(b <= a for a, b in zip(ks, ks[1:])):
        raise ValueError(f"ks must be 1..16 strictly ascending values in [1, {kmax}], got {ks}")
    return (ctypes.c_int32 * len(ks))(*ks), len(ks)


def knn_vote(nbr_idx: torch.Tensor, labels: torch.Tensor, ks: Sequence[int], nclass: int,
             idx_base: int = 0) -> torch.Tensor:
    """pred[j][q] = mode of labels[nbr_idx[q, :ks[j]]], smallest label on ties; int64 [len(ks), nq]."""
    _dev(nbr_idx, "nbr_idx")
    _dev(labels, "labels")
    if nbr_idx.dtype != torch.int64 or labels.dtype != torch.int64 or nbr_idx.dim() != 2 or labels.dim() != 1:
        raise HcirError("knn_vote expects int64 nbr_idx [nq, kmax] and int64 labels [n]")
    nq, kmax = nbr_idx.shape
    arr, nk = _ks(ks, kmax)
    pred = torch.empty((nk, nq), dtype=torch.int64, device=nbr_idx.device)
    bad = torch.zeros(1, dtype=torch.int32, device=nbr_idx.device)
    check(_lib.lib().hcir_knn_vote(nbr_idx.data_ptr(), nq, kmax, int(idx_base), labels.data_ptr(), labels.shape[0],
                                   int(nclass), arr, nk, pred.data_ptr(), bad.data_ptr(), _stream(nbr_idx)),
          "hcir_knn_vote")
    if int(bad.item()):
        raise ValueError("knn_vote: an empty neighbour slot or a label outside [0, nclass)")
    return pred


def confusion_matrix(y_true: torch.Tensor, y_pred: torch.Tensor, nclass: int) -> torch.Tensor:
    _dev(y_true, "y_true")
    _dev(y_pred, "y_pred")
    if y_true.dtype != torch.int64 or y_pred.dtype != torch.int64 or y_true.shape != y_pred.shape:
        raise HcirError("confusion_matrix expects two int64 vectors of equal length")
    cm = torch.zeros((nclass, nclass), dtype=torch.int32, device=y_true.device)
    bad = torch.zeros(1, dtype=torch.int32, device=y_true.device)
    check(_lib.lib().hcir_confusion_matrix(y_true.data_ptr(), y_pred.data_ptr(), y_true.numel(), int(nclass),
                                           cm.data_ptr(), bad.data_ptr(), _stream(y_true)), "hcir_confusion_matrix")
    if int(bad.item()):
        raise ValueError("confusion_matrix: label outside [0, nclass)")
    return cm


def retrieval_metrics(retrieved: torch.Tensor, gt: torch.Tensor, ks: Sequence[int] = (10, 20, 50)
                      ) -> Tuple[Dict[str, Dict[int, float]], torch.Tensor, torch.Tensor]:
    """retrieved int64 [nq, kmax] (ids from the index search), gt int64 [nq, gmax] padded with -1.
    Returns ({'mAP': {k: ..}, 'Recall': {k: ..}, 'total_queries': nq} as the reference's evaluate() does,
    hit int32 [nk, nq], ap float64 [nk, nq])."""
    _dev(retrieved, "retrieved")
    _dev(gt, "gt")
    if retrieved.dtype != torch.int64 or gt.dtype != torch.int64 or retrieved.dim() != 2 or gt.dim() != 2 \
            or retrieved.shape[0] != gt.shape[0]:
        raise HcirError("retrieval_metrics expects int64 retrieved [nq, kmax] and gt [nq, gmax]")
    nq, kmax = retrieved.shape
    arr, nk = _ks(ks, kmax)
    dev = retrieved.device
    hit = torch.empty((nk, nq), dtype=torch.int32, device=dev)
    ap = torch.empty((nk, nq), dtype=torch.float64, device=dev)
    rm = torch.empty(nk, dtype=torch.float64, device=dev)
    mm = torch.empty(nk, dtype=torch.float64, device=dev)
    check(_lib.lib().hcir_retrieval_metrics(retrieved.data_ptr(), nq, kmax, gt.data_ptr(), gt.shape[1], arr, nk,
                                            hit.data_ptr(), ap.data_ptr(), rm.data_ptr(), mm.data_ptr(),
                                            _stream(retrieved)), "hcir_retrieval_metrics")
    rm, mm = rm.cpu().tolist(), mm.cpu().tolist()
    res = {"mAP": {int(k): mm[i] for i, k in enumerate(ks)}, "Recall": {int(k): rm[i] for i, k in enumerate(ks)},
           "total_queries": nq}
    return res, hit, ap
