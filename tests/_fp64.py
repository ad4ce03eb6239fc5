"""fp64 ground truth and a half-operand emulation for the training-path tolerance tests.

  * `Fp64Head`     the projection head (Linear -> BatchNorm1d -> ReLU -> Linear -> BatchNorm1d, train mode) as plain
                   torch modules in float64 — the ground truth;
  * `HalfOperandLinear`  the same Linear with every GEMM operand rounded to fp16 first (forward: x and W; backward:
                   the incoming gradient, W and x), products and sums in float64: what fp16 MFMA operands with exact
                   accumulation would give.  The error of THIS against fp64 is the rounding the operand format
                   forces; a kernel is held to a small multiple of it.
Test infrastructure: runs on whatever device the tensors are on (float64 on the GPU is fine for a ground truth).
"""
import copy

import torch
import torch.nn as nn


def r16(t):
    return t.float().half().to(t.dtype)


class _HalfLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        xq, wq = r16(x), r16(w)
        ctx.save_for_backward(xq, wq)
        return xq @ wq.t()

    @staticmethod
    def backward(ctx, g):
        xq, wq = ctx.saved_tensors
        gq = r16(g)
        return gq @ wq, gq.t() @ xq


class HalfOperandLinear(nn.Module):
    def __init__(self, lin):
        super().__init__()
        self.weight = nn.Parameter(lin.weight.detach().clone())

    def forward(self, x):
        return _HalfLinearFn.apply(x, self.weight)


def fp64_head(head, device, half_operands=False):
    """head.layers (hcir SimCLRProjectionHead) -> an nn.Sequential of torch modules in float64, train mode."""
    lin0, bn0, _, lin1, bn1 = copy.deepcopy(head).cpu().layers
    mods = [HalfOperandLinear(lin0) if half_operands else lin0, bn0, nn.ReLU(),
            HalfOperandLinear(lin1) if half_operands else lin1, bn1]
    return nn.Sequential(*mods).double().to(device).train()


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
