"""hcir.head_train — lightly's SimCLRProjectionHead in TRAINING mode on the HIP path (HP/src/main_backbone.py:589;
called from `self.model(x)` / `forward_momentum` inside the step, HP/src/pretrain_engine.py:683-696).

    Linear(no bias) -> BatchNorm1d (batch statistics) -> ReLU -> Linear(no bias) -> BatchNorm1d (batch statistics)

forward:  hcir_gemm_f16 (fp32 accumulators out) -> hcir_bn1d_fwd (+ReLU, fp16 out) -> hcir_gemm_f16 -> hcir_bn1d_fwd
backward: hcir_bn1d_bwd -> hcir_gemm_f16_tn (dW1) + hcir_gemm_f16 (dgrad) -> hcir_bn1d_bwd (ReLU mask folded in)
          -> hcir_gemm_f16_tn (dW0) + hcir_gemm_f16 (dgrad to the class token)
Running statistics are updated by the forward kernel exactly as torch's BatchNorm1d does (momentum, unbiased variance,
`num_batches_tracked += 1`).  The incoming gradient is renormalised by a power of two before the fp16 operands are
formed (as in hcir.vit_train).  No vendor GEMM, no CPU path.
"""
from __future__ import annotations

import torch

from . import _lib, train_ops as T
from ._lib import HcirError, check


def _pad64(m: int) -> int:
    return (m + 63) // 64 * 64


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _gemm(L, a16, k, w16, m, n, out32, st, what):
    """out32[m, n] (fp32) = a16[m, k] . w16[n, k]^T"""
    check(L.hcir_gemm_f16(a16.data_ptr(), k, w16.data_ptr(), k, None, None, m, n, k, _lib.EPI_BIAS_F32,
                          out32.data_ptr(), n, st), what)


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, g0, b0, w1, g1, b1, bn0, bn1):
        if not x.is_cuda:
            raise HcirError(f"projection head input is on {x.device}; the training head runs on a HIP device only")
        L = _lib.lib()
        dev = x.device
        st = _st(dev)
        m, k0 = x.shape
        hid, out_dim = w0.shape[0], w1.shape[0]
        if m < 2:
            raise ValueError("Expected more than 1 value per channel when training")   # torch's BatchNorm check
        mp = _pad64(m)
        x16 = torch.zeros((mp, k0), dtype=torch.float16, device=dev)
        x16[:m] = x.detach()
        w0h, w1h = w0.detach().half().contiguous(), w1.detach().half().contiguous()
        h0 = torch.empty((m, hid), dtype=torch.float32, device=dev)
        _gemm(L, x16, k0, w0h, m, hid, h0, st, "hcir_gemm_f16(head.0)")
        a0 = torch.zeros((mp, hid), dtype=torch.float16, device=dev)
        mean0, rstd0 = (torch.empty(hid, dtype=torch.float32, device=dev) for _ in range(2))

        def bn_fwd(bn, h, f, gamma, beta, relu, mean, rstd, y32, y16):
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            track = bn.track_running_stats and bn.running_mean is not None
            check(L.hcir_bn1d_fwd(h.data_ptr(), f, m, f, gamma.detach().float().contiguous().data_ptr(),
                                  beta.detach().float().contiguous().data_ptr(), float(bn.eps), mom, int(relu),
                                  bn.running_mean.data_ptr() if track else None,
                                  bn.running_var.data_ptr() if track else None, mean.data_ptr(), rstd.data_ptr(),
                                  None if y32 is None else y32.data_ptr(), f, None if y16 is None else y16.data_ptr(), f,
                                  st), "hcir_bn1d_fwd")
            if track and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1

        bn_fwd(bn0, h0, hid, g0, b0, True, mean0, rstd0, None, a0)
        h1 = torch.empty((m, out_dim), dtype=torch.float32, device=dev)
        _gemm(L, a0, hid, w1h, m, out_dim, h1, st, "hcir_gemm_f16(head.3)")
        mean1, rstd1 = (torch.empty(out_dim, dtype=torch.float32, device=dev) for _ in range(2))
        out = torch.empty((m, out_dim), dtype=torch.float32, device=dev)
        bn_fwd(bn1, h1, out_dim, g1, b1, False, mean1, rstd1, out, None)
        ctx.save_for_backward(x16, w0h, w1h, h0, a0, h1, mean0, rstd0, mean1, rstd1, g0.detach().float().contiguous(),
                              g1.detach().float().contiguous())
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, dout):
        x16, w0h, w1h, h0, a0, h1, mean0, rstd0, mean1, rstd1, g0, g1 = ctx.saved_tensors
        L = _lib.lib()
        dev = dout.device
        st = _st(dev)
        m, mp = ctx.m, x16.shape[0]
        k0, hid, out_dim = x16.shape[1], w0h.shape[0], w1h.shape[0]
        dout = dout.float().contiguous()
        # power-of-two renormalisation of the incoming gradient (fp16 operands behind it), undone on the results
        amax = dout.abs().max()
        ok = torch.isfinite(amax) & (amax > 0)
        kexp = torch.where(ok, -torch.floor(torch.log2(torch.where(ok, amax, torch.ones_like(amax)))) - 1.0,
                           torch.zeros_like(amax)).clamp_(-100.0, 100.0)
        up, down = torch.exp2(kexp).float().reshape(1), torch.exp2(-kexp).float()
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        # BN1 backward
        dh1 = torch.zeros((mp, out_dim), dtype=torch.float16, device=dev)
        dg1, db1 = f32(out_dim), f32(out_dim)
        check(L.hcir_bn1d_bwd(dout.data_ptr(), out_dim, h1.data_ptr(), out_dim, m, out_dim, g1.data_ptr(), mean1.data_ptr(),
                              rstd1.data_ptr(), None, 0, up.data_ptr(), dh1.data_ptr(), out_dim, dg1.data_ptr(),
                              db1.data_ptr(), st), "hcir_bn1d_bwd(4)")
        # Linear 1: dW1 = dh1^T a0, da0 = dh1 W1
        dw1 = f32(out_dim, hid)
        T.gemm_tn(dh1, a0, dw1, accumulate=False)
        da0 = f32(m, hid)
        _gemm(L, dh1, out_dim, w1h.t().contiguous(), m, hid, da0, st, "hcir_gemm_f16(dgrad head.3)")
        # ReLU + BN0 backward (the ReLU mask is read from the forward's fp16 output)
        dh0 = torch.zeros((mp, hid), dtype=torch.float16, device=dev)
        dg0, db0 = f32(hid), f32(hid)
        check(L.hcir_bn1d_bwd(da0.data_ptr(), hid, h0.data_ptr(), hid, m, hid, g0.data_ptr(), mean0.data_ptr(),
                              rstd0.data_ptr(), a0.data_ptr(), hid, None, dh0.data_ptr(), hid, dg0.data_ptr(),
                              db0.data_ptr(), st), "hcir_bn1d_bwd(1)")
        dw0 = f32(hid, k0)
        T.gemm_tn(dh0, x16, dw0, accumulate=False)
        dx = f32(m, k0)
        _gemm(L, dh0, hid, w0h.t().contiguous(), m, k0, dx, st, "hcir_gemm_f16(dgrad head.0)")
        grads = [dx, dw0, dg0, db0, dw1, dg1, db1]
        torch._foreach_mul_(grads, down)
        return tuple(grads) + (None, None)


def head_train_forward(head, x: torch.Tensor) -> torch.Tensor:
    """SimCLRProjectionHead.forward in train mode on the HIP path (differentiable; also runs under no_grad for the
    momentum head, whose BatchNorm layers still take batch statistics and update their running ones)."""
    lin0, bn0, _, lin1, bn1 = head.layers
    if lin0.bias is not None or lin1.bias is not None:
        raise HcirError("SimCLRProjectionHead's Linear layers carry no bias")
    for f in (lin0.weight.shape[0], lin0.weight.shape[1], lin1.weight.shape[0]):
        if f % 256:
            raise HcirError("the HIP training head needs feature sizes that are multiples of 256 (hcir_gemm_f16_tn)")
    return _HeadFn.apply(x.float(), lin0.weight, bn0.weight, bn0.bias, lin1.weight, bn1.weight, bn1.bias, bn0, bn1)
