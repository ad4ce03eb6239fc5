"""hcir.gallery — a gallery resident in HBM and the "exact by verification" search.

`ResidentGallery(g32)` keeps the fp32 rows (the reference's gallery) and, optionally, an fp16
MIRROR of them.  `search(q, k)` returns exactly what the exact fp32 scan returns — same indices,
bit-identical scores — but streams only the half-size mirror for (almost) every query:

  1. filter   hcir_sim_topk(fp16 q, fp16 mirror, kc = 16)            HBM-streaming / fp16 MFMA
  2. refine   hcir_topk_refine_f32: re-score the kc candidates with the exact fp32 fmaf chain,
              rank them, certify  s16[kc-1] + E_i < exact k-th score
  3. fallback uncertified queries (rare) go through hcir_sim_topk(HCIR_F32) on the fp32 rows.

Why step 2 proves exactness.  Let q~, g~ be the fp16 roundings.  For every gallery row j
    |chain32(q_i, g_j) - s16(q~_i, g~_j)|
        <= |q.g - q~.g~| + |chain32 - q.g| + |s16 - q~.g~|
        <= ||q_i|| * max_j ||g_j - g~_j||  +  ||q_i - q~_i|| * max_j ||g~_j||          (Cauchy-Schwarz)
           + gamma_d * ||q_i|| max_j ||g_j||  +  gamma_d * ||q~_i|| max_j ||g~_j||     (any fp32 summation
             order, one rounding per term; products of two fp16 are exact in fp32; gamma_d = d u/(1 - d u), u = 2^-24)
        =: E_i.
The rounding-error norms are MEASURED (exactly, in fp32 on the device) when the mirror is built and
per query batch, not worst-cased at 2^-11 per element, which keeps E_i ~5e-4 for unit-norm rows.
A row outside the candidate set has s16 <= s16[kc-1], hence chain32 <= s16[kc-1] + E_i; if that is
below the exact k-th candidate score the row cannot be in the exact top-k (ties included: the
inequality is strict).  A 1.01 safety factor covers the fp32 evaluation of E_i itself.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib, ops
from ._lib import HcirError, check

FILTER_KC = 16  # candidates per query from the mirror scan (the KP = 16 register-list kernel)


class ResidentGallery:
    def __init__(self, g32: torch.Tensor, idx_base: int = 0, mirror: Optional[torch.dtype] = torch.float16):
        ops._dev(g32, "gallery")
        if g32.dtype != torch.float32 or g32.dim() != 2:
            raise HcirError("ResidentGallery expects an fp32 [N, D] tensor")
        self.g32 = g32
        self.idx_base = int(idx_base)
        self.mirror = None
        self.stats = {"calls": 0, "queries": 0, "fallback_queries": 0}
        if mirror is not None:
            self.mirror = ops.convert(g32, mirror)
            n, d = g32.shape
            gmax32 = gmax16 = egmax = 0.0
            for s in range(0, n, 262144):  # chunked: the fp32 difference is a temporary
                e = min(n, s + 262144)
                m32 = self.mirror[s:e].float()
                egmax = max(egmax, float((g32[s:e] - m32).norm(dim=1).max()))
                gmax16 = max(gmax16, float(m32.norm(dim=1).max()))
                gmax32 = max(gmax32, float(g32[s:e].norm(dim=1).max()))
            u = 2.0 ** -24
            gamma = d * u / (1.0 - d * u)
            self._eg, self._g16max, self._g32max, self._gamma = egmax, gmax16, gmax32, gamma
            import ctypes
            self._consts = (ctypes.c_float * 4)(egmax, gmax16, gmax32, gamma)  # host array for the refine kernel

    # ---- exact fp32 scan ---------------------------------------------------------------------
    def search_exact(self, q32: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return ops.sim_topk(q32, self.g32, k, idx_base=self.idx_base)

    # ---- filter + refine + fallback ----------------------------------------------------------
    def err_bound(self, q32: torch.Tensor, q16: torch.Tensor) -> torch.Tensor:
        """E_i of the module docstring, fp32 [nq] on the device (no host sync)."""
        qf = q16.float()
        qn = q32.norm(dim=1)
        e = qn * self._eg + (q32 - qf).norm(dim=1) * self._g16max \
            + self._gamma * (qn * self._g32max + qf.norm(dim=1) * self._g16max)
        return (e * 1.01 + 1e-7).contiguous()

    def search_begin(self, q32: torch.Tensor, k: int, q16: Optional[torch.Tensor] = None) -> "PendingSearch":
        """Launch filter + refine (asynchronous, no host sync) and return a handle; `finish()` checks
        the certification flags (one host sync) and re-runs uncertified queries through the exact scan.
        Splitting the two lets a caller enqueue its NEXT batch's work before it blocks on the flags,
        so the GPU never idles behind the check (bench.py does exactly that)."""
        if self.mirror is None or k > FILTER_KC - 2 or self.g32.shape[0] <= FILTER_KC:
            val, idx = self.search_exact(q32, k)
            return PendingSearch(self, q32, k, val, idx, None)
        ops._dev(q32, "q")
        nq, d = q32.shape
        if q16 is None or q16.dtype != self.mirror.dtype or q16.shape != q32.shape:
            q16 = q32.to(self.mirror.dtype)  # callers that already hold the rounded copy (forward_cls(want_f16)) pass it
        cval, cidx = ops.sim_topk(q16, self.mirror, FILTER_KC, idx_base=self.idx_base)
        val = torch.empty((nq, k), dtype=torch.float32, device=q32.device)
        idx = torch.empty((nq, k), dtype=torch.int64, device=q32.device)
        cert = torch.empty(nq, dtype=torch.int32, device=q32.device)
        check(_lib.lib().hcir_topk_refine_f32(
            q32.data_ptr(), nq, self.g32.data_ptr(), self.g32.shape[0], d, cidx.data_ptr(), cval.data_ptr(),
            FILTER_KC, k, self.idx_base, None, None, None, self._consts, val.data_ptr(), idx.data_ptr(),
            cert.data_ptr(), ops._stream(q32)), "hcir_topk_refine_f32")
        return PendingSearch(self, q32, k, val, idx, cert)

    def search(self, q32: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Exact fp32 top-k of q32 against the gallery (identical to search_exact)."""
        return self.search_begin(q32, k).finish()


class PendingSearch:
    """Result of ResidentGallery.search_begin: device tensors already enqueued; finish() certifies."""

    def __init__(self, gallery, q32, k, val, idx, cert):
        self.gallery, self.q32, self.k, self.val, self.idx, self.cert = gallery, q32, k, val, idx, cert

    def finish(self) -> Tuple[torch.Tensor, torch.Tensor]:
        g = self.gallery
        if self.cert is not None:
            # the one host sync of a batch: a single reduction; the index list (four more launches) only when needed
            all_ok = bool(self.cert.min().item())
            g.stats["calls"] += 1
            g.stats["queries"] += self.q32.shape[0]
            bad = None if all_ok else (self.cert == 0).nonzero().flatten()
            if bad is not None and bad.numel():
                g.stats["fallback_queries"] += int(bad.numel())
                bv, bi = g.search_exact(self.q32[bad].contiguous(), self.k)
                self.val[bad] = bv
                self.idx[bad] = bi
            self.cert = None
        return self.val, self.idx
