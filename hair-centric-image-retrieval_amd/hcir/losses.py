"""hcir.losses — NTXentLoss with lightly's constructor/forward signature
(lightly.loss.NTXentLoss; call sites HP/src/pretrain_engine.py:93,725), computed by
hcir_ntxent_fwd (fused normalise + 2B x 2B MFMA cosine + masked log-sum-exp) and, when the inputs
require grad, differentiated by hcir_ntxent_bwd through a torch.autograd.Function — a drop-in
`criterion(out0, out1)` for a training loop whose backbone runs on PyTorch-ROCm.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from ._lib import HcirError, check
from .ops import _DT, _dev, _stream, _ws


def ntxent_forward(out0: torch.Tensor, out1: torch.Tensor, temperature: float, want_lse: bool = False):
    _dev(out0, "out0")
    _dev(out1, "out1")
    if out0.shape != out1.shape or out0.dim() != 2 or out0.dtype != out1.dtype:
        raise HcirError(f"NTXentLoss expects two [B, D] tensors of one dtype, got {tuple(out0.shape)} "
                        f"{out0.dtype} / {tuple(out1.shape)} {out1.dtype}")
    if out0.dtype not in _DT:
        raise HcirError(f"unsupported dtype {out0.dtype}")
    b, d = out0.shape
    L = _lib.lib()
    dt = _DT[out0.dtype]
    ws = _ws.get(out0.device, L.hcir_ntxent_workspace_bytes(b, d, dt))
    loss = torch.empty((), dtype=torch.float32, device=out0.device)
    lse = torch.empty(2 * b, dtype=torch.float32, device=out0.device) if want_lse else None
    check(L.hcir_ntxent_fwd(out0.data_ptr(), out1.data_ptr(), b, d, dt, 1.0 / temperature,
                            loss.data_ptr(), None if lse is None else lse.data_ptr(), ws.data_ptr(),
                            ws.numel(), _stream(out0)), "hcir_ntxent_fwd")
    return (loss, lse) if want_lse else loss


def ntxent_backward(out0, out1, temperature, lse, grad_out: float):
    b, d = out0.shape
    L = _lib.lib()
    dt = _DT[out0.dtype]
    ws = _ws.get(out0.device, L.hcir_ntxent_bwd_workspace_bytes(b, d, dt))
    g0, g1 = torch.empty_like(out0), torch.empty_like(out1)
    check(L.hcir_ntxent_bwd(out0.data_ptr(), out1.data_ptr(), b, d, dt, 1.0 / temperature, lse.data_ptr(),
                            float(grad_out), g0.data_ptr(), g1.data_ptr(), ws.data_ptr(), ws.numel(),
                            _stream(out0)), "hcir_ntxent_bwd")
    return g0, g1


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out0, out1, temperature):
        out0, out1 = out0.contiguous(), out1.contiguous()
        loss, lse = ntxent_forward(out0, out1, temperature, want_lse=True)
        ctx.save_for_backward(out0, out1, lse)
        ctx.temperature = temperature
        return loss

    @staticmethod
    def backward(ctx, grad):
        out0, out1, lse = ctx.saved_tensors
        g0, g1 = ntxent_backward(out0, out1, ctx.temperature, lse, float(grad))  # one host read of dL
        return g0, g1, None


class NTXentLoss(nn.Module):
    """NTXentLoss(temperature=0.5, memory_bank_size=0, gather_distributed=False)."""

    def __init__(self, temperature: float = 0.5, memory_bank_size=0, gather_distributed: bool = False):
        super().__init__()
        if abs(temperature) < 1e-8:
            raise ValueError("Illegal temperature: abs({}) < 1e-8".format(temperature))
        if memory_bank_size not in (0, (0, 0)):
            raise NotImplementedError("memory bank is never enabled by the reference "
                                      "(HP/src/pretrain_engine.py:74,93)")
        if gather_distributed:
            raise NotImplementedError("gather_distributed is never enabled by the reference "
                                      "(SURVEY.md §2.3)")
        self.temperature = temperature

    def forward(self, out0: torch.Tensor, out1: torch.Tensor) -> torch.Tensor:
        if torch.is_grad_enabled() and (out0.requires_grad or out1.requires_grad):
            if out0.shape[0] % 4:
                raise HcirError("hcir_ntxent_bwd needs a batch size that is a multiple of 4")
            return _NTXentFn.apply(out0, out1, self.temperature)
        return ntxent_forward(out0.contiguous(), out1.contiguous(), self.temperature)
