"""hcir.train_ops — torch-tensor wrappers over the training half of the C ABI (include/hcir.h, "Training side").

Backward building blocks of the ViT and the two small losses of the HSimCLR step
(HP/src/pretrain_engine.py:681-751).  Every wrapper takes HIP-device tensors and raises otherwise: no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import HcirError, check
from .ops import _dev, _stream, _ws

_XDT = {torch.float32: _lib.F32, torch.float16: _lib.F16}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def gelu_fwd(u: torch.Tensor) -> torch.Tensor:
    _dev(u, "u")
    if u.dtype != torch.float16 or u.numel() % 8:
        raise HcirError("gelu_fwd expects a contiguous fp16 tensor with numel % 8 == 0")
    h = torch.empty_like(u)
    check(_lib.lib().hcir_gelu_fwd_f16(u.data_ptr(), u.numel(), h.data_ptr(), _stream(u)), "hcir_gelu_fwd_f16")
    return h


def gelu_bwd(u: torch.Tensor, dh: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(u, "u")
    _dev(dh, "dh")
    if u.dtype != torch.float16 or dh.dtype != torch.float16 or u.shape != dh.shape or u.numel() % 8:
        raise HcirError("gelu_bwd expects two contiguous fp16 tensors of equal shape, numel % 8 == 0")
    du = torch.empty_like(u) if out is None else out
    check(_lib.lib().hcir_gelu_bwd_f16(u.data_ptr(), dh.data_ptr(), u.numel(), du.data_ptr(), _stream(u)),
          "hcir_gelu_bwd_f16")
    return du


def add_to_f16(a: torch.Tensor, b: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp16(a + b) of fp32 tensors (b optional)."""
    _dev(a, "a")
    if a.dtype != torch.float32 or a.numel() % 4:
        raise HcirError("add_to_f16 expects contiguous fp32, numel % 4 == 0")
    y = torch.empty(a.shape, dtype=torch.float16, device=a.device) if out is None else out
    check(_lib.lib().hcir_add_f32_f16(a.data_ptr(), _p(b), a.numel(), y.data_ptr(), _stream(a)), "hcir_add_f32_f16")
    return y


def layernorm_bwd(x: torch.Tensor, dy16: torch.Tensor, gamma: torch.Tensor, eps: float,
                  dres_in: Optional[torch.Tensor], dres_out: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
                  accumulate: bool = True, rows: Optional[int] = None, ldx: Optional[int] = None,
                  ldr: Optional[int] = None, dres16: Optional[torch.Tensor] = None,
                  dres_colsum: Optional[torch.Tensor] = None) -> None:
    """dres_out = dres_in + LayerNorm'(x)[dy16];  dgamma / dbeta (+)= the column reductions.
    x [rows, d] (fp32 or fp16, row pitch ldx), dy16 fp16 [rows, d], dres_* fp32 (row pitch ldr).
    dres16 (fp16 [rows, d]): also receives fp16(dres_out); dres_colsum (fp32 [d]): the column sums of that copy."""
    for t, n in ((x, "x"), (dy16, "dy"), (gamma, "gamma"), (dres_out, "dres_out"), (dgamma, "dgamma"), (dbeta, "dbeta")):
        _dev(t, n)
    d = gamma.numel()
    rows = dy16.shape[0] if rows is None else rows
    ldx = d if ldx is None else ldx
    ldr = d if ldr is None else ldr
    L = _lib.lib()
    nb = L.hcir_layernorm_bwd_blocks(rows)
    ws = _ws.get(x.device, nb * d * 12)
    if dres16 is not None:
        _dev(dres16, "dres16")
        if dres16.dtype != torch.float16 or dres16.stride(0) != d:
            raise HcirError("dres16: fp16 [rows, d], contiguous rows")
    check(L.hcir_layernorm_bwd_fused(x.data_ptr(), _XDT[x.dtype], rows, d, ldx, dy16.data_ptr(), d, gamma.data_ptr(),
                                     float(eps), _p(dres_in), dres_out.data_ptr(), ldr, dgamma.data_ptr(),
                                     dbeta.data_ptr(), int(accumulate), _p(dres16), d, _p(dres_colsum), ws.data_ptr(),
                                     ws.numel(), _stream(x)), "hcir_layernorm_bwd_fused")


def gelu_bwd_colsum(u: torch.Tensor, dh: torch.Tensor, du: torch.Tensor, colsum_out: torch.Tensor,
                    rows: Optional[int] = None, accumulate: bool = False) -> None:
    """du = dh * gelu'(u) over the first `rows` rows of [m, n] fp16 matrices, and colsum_out[n] (+)= sum_rows du."""
    for t, nm in ((u, "u"), (dh, "dh"), (du, "du"), (colsum_out, "colsum")):
        _dev(t, nm)
    if u.dtype != torch.float16 or dh.dtype != torch.float16 or du.dtype != torch.float16 or u.shape != dh.shape:
        raise HcirError("gelu_bwd_colsum expects fp16 matrices of equal shape")
    m = u.shape[0] if rows is None else rows
    n = u.shape[1]
    L = _lib.lib()
    ws = _ws.get(u.device, L.hcir_colsum_chunks(m) * n * 4)
    check(L.hcir_gelu_bwd_colsum_f16(u.data_ptr(), dh.data_ptr(), m, n, u.stride(0), du.data_ptr(),
                                     colsum_out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel(), _stream(u)),
          "hcir_gelu_bwd_colsum_f16")


def colsum(x16: torch.Tensor, out: torch.Tensor, accumulate: bool = True, rows: Optional[int] = None) -> None:
    """out[n] (+)= sum over rows of the fp16 matrix x16 [m, n]."""
    _dev(x16, "x")
    _dev(out, "out")
    m = x16.shape[0] if rows is None else rows
    n = x16.shape[1]
    L = _lib.lib()
    ws = _ws.get(x16.device, L.hcir_colsum_chunks(m) * n * 4)
    check(L.hcir_colsum_f16(x16.data_ptr(), m, n, x16.stride(0), out.data_ptr(), int(accumulate), ws.data_ptr(),
                            ws.numel(), _stream(x16)), "hcir_colsum_f16")


def gemm_tn(a16: torch.Tensor, b16: torch.Tensor, dw: torch.Tensor, accumulate: bool = True) -> None:
    """dw[N, K] (+)= a16[M, N]^T @ b16[M, K]   (fp16 operands, fp32 result).  M % 64 == 0 (the caller pads with zero
    rows), N % 256 == 0, K % 256 == 0."""
    for t, n in ((a16, "a"), (b16, "b"), (dw, "dw")):
        _dev(t, n)
    if a16.dtype != torch.float16 or b16.dtype != torch.float16 or dw.dtype != torch.float32:
        raise HcirError("gemm_tn: fp16 operands, fp32 output")
    m, n = a16.shape
    k = b16.shape[1]
    if b16.shape[0] != m or tuple(dw.shape) != (n, k):
        raise HcirError(f"gemm_tn shape mismatch: a {tuple(a16.shape)} b {tuple(b16.shape)} dw {tuple(dw.shape)}")
    L = _lib.lib()
    wsb = L.hcir_gemm_f16_tn_workspace_bytes(m, n, k)
    if wsb == 0:
        raise HcirError(f"gemm_tn unsupported shape M={m} N={n} K={k} (M % 64, N % 256, K % 256)")
    ws = _ws.get(a16.device, wsb)
    check(L.hcir_gemm_f16_tn(a16.data_ptr(), a16.stride(0), b16.data_ptr(), b16.stride(0), m, n, k, dw.data_ptr(),
                             dw.stride(0), int(accumulate), ws.data_ptr(), ws.numel(), _stream(a16)),
          "hcir_gemm_f16_tn")


def attn_fwd_lse(qkv: torch.Tensor, b: int, t: int, heads: int, scale: float, out: torch.Tensor,
                 lse: torch.Tensor) -> None:
    check(_lib.lib().hcir_attn_fwd_lse(qkv.data_ptr(), b, t, heads, 64, float(scale), out.data_ptr(), lse.data_ptr(),
                                       _stream(qkv)), "hcir_attn_fwd_lse")


def attn_bwd(qkv: torch.Tensor, out: torch.Tensor, d_out: torch.Tensor, lse: torch.Tensor, b: int, t: int, heads: int,
             scale: float, d_qkv: torch.Tensor) -> None:
    check(_lib.lib().hcir_attn_bwd(qkv.data_ptr(), out.data_ptr(), d_out.data_ptr(), lse.data_ptr(), b, t, heads, 64,
                                   float(scale), d_qkv.data_ptr(), _stream(qkv)), "hcir_attn_bwd")


# ---------------------------------------------------------------------------------------------------------------
# losses (torch.autograd Functions over the HIP kernels)
# ---------------------------------------------------------------------------------------------------------------
class _TripletFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, n, margin, eps):
        a, p, n = (t.detach().float().contiguous() for t in (a, p, n))
        for t, nm in ((a, "anchor"), (p, "positive"), (n, "negative")):
            _dev(t, nm)
        b, d = a.shape
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        row = torch.empty(b, dtype=torch.float32, device=a.device)
        dist = torch.empty((2, b), dtype=torch.float32, device=a.device)
        check(_lib.lib().hcir_triplet_margin_fwd(a.data_ptr(), p.data_ptr(), n.data_ptr(), b, d, float(margin),
                                                 float(eps), loss.data_ptr(), row.data_ptr(), dist.data_ptr(),
                                                 _stream(a)), "hcir_triplet_margin_fwd")
        ctx.save_for_backward(a, p, n, row, dist)
        ctx.eps = float(eps)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, p, n, row, dist = ctx.saved_tensors
        b, d = a.shape
        g = g.detach().float().contiguous().reshape(1)
        da, dp, dn = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        check(_lib.lib().hcir_triplet_margin_bwd(a.data_ptr(), p.data_ptr(), n.data_ptr(), b, d, ctx.eps,
                                                 row.data_ptr(), dist.data_ptr(), g.data_ptr(), da.data_ptr(),
                                                 dp.data_ptr(), dn.data_ptr(), _stream(a)), "hcir_triplet_margin_bwd")
        return da, dp, dn, None, None


class TripletMarginLoss(torch.nn.Module):
    """nn.TripletMarginLoss(margin, p=2, eps, swap=False, reduction='mean') on the HIP path
    (HP/src/pretrain_engine.py:96-97)."""

    def __init__(self, margin: float = 1.0, p: float = 2.0, eps: float = 1e-6, swap: bool = False,
                 reduction: str = "mean"):
        super().__init__()
        if p != 2 or swap or reduction != "mean":
            raise NotImplementedError("hcir TripletMarginLoss implements p=2, swap=False, reduction='mean'")
        self.margin, self.p, self.eps = margin, p, eps

    def forward(self, anchor, positive, negative):
        return _TripletFn.apply(anchor, positive, negative, self.margin, self.eps)


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = x.detach().float().contiguous(), y.detach().float().contiguous()
        _dev(x, "input")
        _dev(y, "target")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        ws = torch.empty(256, dtype=torch.float32, device=x.device)
        check(_lib.lib().hcir_mse_fwd(x.data_ptr(), y.data_ptr(), x.numel(), loss.data_ptr(), ws.data_ptr(),
                                      _stream(x)), "hcir_mse_fwd")
        ctx.save_for_backward(x, y)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.detach().float().contiguous().reshape(1)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if dx is None and dy is None:
            return None, None
        check(_lib.lib().hcir_mse_bwd(x.data_ptr(), y.data_ptr(), x.numel(), g.data_ptr(), _p(dx), _p(dy), _stream(x)),
              "hcir_mse_bwd")
        return dx, dy


def mse_loss(input: torch.Tensor, target: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """F.mse_loss(input, target, reduction='mean') on the HIP path (HP/src/pretrain_engine.py:730)."""
    if reduction != "mean":
        raise NotImplementedError("hcir mse_loss implements reduction='mean'")
    if input.shape != target.shape:
        raise ValueError("mse_loss: shapes differ")
    return _MseFn.apply(input, target)
