"""hcir.ops — thin torch-tensor wrappers over the C ABI (include/hcir.h).

Every wrapper takes HIP-device tensors, passes raw device pointers and the current
stream, and raises if the tensor lives anywhere else: no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import BF16, F16, F32, HcirError, check

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


def _dev(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise HcirError(f"hcir op got `{name}` on {t.device}; the hot path runs on a HIP device "
                        "only (no CPU fallback)")
    if not t.is_contiguous():
        raise HcirError(f"`{name}` must be contiguous")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def row_invnorm(x: torch.Tensor, eps: float) -> torch.Tensor:
    """1 / max(||x_i||, eps) per row (fp32)."""
    _dev(x, "x")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.lib().hcir_row_invnorm(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
                                      _DT[x.dtype], eps, out.data_ptr(), _stream(x)), "hcir_row_invnorm")
    return out


def l2_normalize(x: torch.Tensor, eps: float = 1e-12, want_f16: bool = False):
    """F.normalize(x, dim=1) for fp32 rows; optionally also an fp16 copy."""
    _dev(x, "x")
    if x.dtype != torch.float32:
        raise HcirError("l2_normalize expects fp32")
    y = torch.empty_like(x)
    y16 = torch.empty(x.shape, dtype=torch.float16, device=x.device) if want_f16 else None
    check(_lib.lib().hcir_l2_normalize(x.data_ptr(), x.shape[0], x.shape[1], eps, y.data_ptr(),
                                       _ptr(y16), _stream(x)), "hcir_l2_normalize")
    return (y, y16) if want_f16 else y


def convert(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 -> fp16/bf16 copy (gallery upload)."""
    _dev(x, "x")
    if x.dtype != torch.float32:
        raise HcirError("convert expects fp32 input")
    if dtype == torch.float32:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(_lib.lib().hcir_convert_f32(x.data_ptr(), x.numel(), _DT[dtype], y.data_ptr(), _stream(x)),
          "hcir_convert_f32")
    return y


class _WorkspaceCache:
    """Scratch buffers for the C ABI (which allocates nothing itself), one per (device, stream): two streams
    never share a buffer, so concurrent calls cannot race on it.  A buffer grows on demand and is dropped only after
    SHRINK_AFTER consecutive requests that would fit an eighth of it (a one-off k = 642 sweep does not pin its
    gigabytes for the process lifetime, while a training backward that alternates 66 MB weight-gradient workspaces
    with KB-sized reductions keeps ONE buffer instead of re-allocating it several hundred times per step)."""

    SHRINK_AFTER = 64

    def __init__(self):
        self._buf = {}
        self._small = {}

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        b = self._buf.get(key)
        if b is not None and b.numel() >= nbytes:
            if b.numel() > max(8 * nbytes, 64 << 20):
                self._small[key] = self._small.get(key, 0) + 1
                if self._small[key] < self.SHRINK_AFTER:
                    return b
            else:
                self._small[key] = 0
                return b
        b = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        self._buf[key] = b
        self._small[key] = 0
        return b


_ws = _WorkspaceCache()
WORKSPACE_LIMIT_BYTES = 1 << 30   # sim_topk splits its queries into chunks whose workspace stays below this


def sim_topk(q: torch.Tensor, g: torch.Tensor, k: int, q_inv_norm: Optional[torch.Tensor] = None,
             g_inv_norm: Optional[torch.Tensor] = None, idx_base: int = 0
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k of (<q_i, g_j> * g_inv_norm[j]) * q_inv_norm[i]; (values fp32, indices int64).

    Ties: score descending, then index ascending.  Raises ValueError when k > len(g),
    as sklearn's kneighbors does (HP/src/classification_engine.py:71 sweeps k up to 642).
    """
    _dev(q, "q")
    _dev(g, "g")
    if q.dtype != g.dtype or q.dtype not in _DT:
        raise HcirError(f"q/g dtypes must match and be fp32/fp16/bf16, got {q.dtype}/{g.dtype}")
    if q.dim() != 2 or g.dim() != 2 or q.shape[1] != g.shape[1]:
        raise HcirError(f"shape mismatch: q {tuple(q.shape)} g {tuple(g.shape)}")
    nq, d = q.shape
    ng = g.shape[0]
    if k > ng:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, "
                         f"n_samples_fit = {ng}, n_samples = {nq}")
    if k < 1:
        raise ValueError(f"Expected n_neighbors > 0. Got {k}")
    for t, n in ((q_inv_norm, "q_inv_norm"), (g_inv_norm, "g_inv_norm")):
        if t is not None:
            _dev(t, n)
            if t.dtype != torch.float32:
                raise HcirError(f"{n} must be fp32")
    L = _lib.lib()
    val = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    # bounded workspace: the partial lists are 2 * 512 * nq * {16,32,64} * 4 B (270 KB per query at k > 32), so a
    # large query set is scanned in chunks (queries are independent; each chunk re-streams the gallery)
    chunk = nq
    while chunk > 128 and L.hcir_sim_topk_workspace_bytes(chunk, ng, d, k, _DT[q.dtype]) > WORKSPACE_LIMIT_BYTES:
        chunk = (chunk + 1) // 2
    if chunk < nq:
        chunk = (chunk + 127) // 128 * 128
    for s0 in range(0, nq, chunk):
        n = min(chunk, nq - s0)
        wsb = L.hcir_sim_topk_workspace_bytes(n, ng, d, k, _DT[q.dtype])
        ws = _ws.get(q.device, wsb)
        qn = None if q_inv_norm is None else q_inv_norm[s0:s0 + n]
        check(L.hcir_sim_topk(q[s0:s0 + n].data_ptr(), n, g.data_ptr(), ng, d, k, _DT[q.dtype], _ptr(qn),
                              _ptr(g_inv_norm), idx_base, val[s0:s0 + n].data_ptr(), idx[s0:s0 + n].data_ptr(),
                              ws.data_ptr(), ws.numel(), _stream(q)), "hcir_sim_topk")
    return val, idx


def topk_merge(vals: torch.Tensor, idx: torch.Tensor, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge [nlists, nq, k_in] sorted lists into [nq, k_out]."""
    _dev(vals, "vals")
    _dev(idx, "idx")
    if vals.dtype != torch.float32 or idx.dtype != torch.int64 or vals.shape != idx.shape:
        raise HcirError("topk_merge expects fp32 values and int64 indices of equal shape")
    nl, nq, kin = vals.shape
    ov = torch.empty((nq, k_out), dtype=torch.float32, device=vals.device)
    oi = torch.empty((nq, k_out), dtype=torch.int64, device=vals.device)
    check(_lib.lib().hcir_topk_merge(vals.data_ptr(), idx.data_ptr(), nl, nq, kin, k_out,
                                     ov.data_ptr(), oi.data_ptr(), _stream(vals)), "hcir_topk_merge")
    return ov, oi
