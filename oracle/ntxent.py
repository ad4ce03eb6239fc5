"""oracle.ntxent — NT-Xent restatements (TEST INFRASTRUCTURE ONLY).

lightly.loss.NTXentLoss (call sites HP/src/pretrain_engine.py:93,725) is not
installed; its 4-block formulation is restated from SURVEY.md Appendix A.  The
reference's own in-tree equivalent experiments/DualViewHair/src/losses/ntxent_loss.py
IS importable in the build container and generated tests/golden/ntxent_*.npz
(tests/golden/make_golden.py); tests/test_oracle_ntxent.py pins both restatements to
those vectors.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def ntxent_lightly(out0: torch.Tensor, out1: torch.Tensor, temperature: float = 0.5) -> torch.Tensor:
    """lightly NTXentLoss.forward, memory_bank_size=0, gather_distributed=False."""
    if abs(temperature) < 1e-8:
        raise ValueError(f"Illegal temperature: abs({temperature}) < 1e-8")
    b = out0.shape[0]
    out0 = F.normalize(out0, dim=1)
    out1 = F.normalize(out1, dim=1)
    l00 = out0 @ out0.t() / temperature
    l01 = out0 @ out1.t() / temperature
    l10 = out1 @ out0.t() / temperature
    l11 = out1 @ out1.t() / temperature
    off = ~torch.eye(b, dtype=torch.bool)
    l00 = l00[off].view(b, -1)
    l11 = l11[off].view(b, -1)
    logits = torch.cat([torch.cat([l01, l00], 1), torch.cat([l10, l11], 1)], 0)
    labels = torch.arange(b).repeat(2)
    return F.cross_entropy(logits, labels)


def ntxent_dualview(z0: torch.Tensor, z1: torch.Tensor, temperature: float) -> torch.Tensor:
    """experiments/DualViewHair/src/losses/ntxent_loss.py:30-57 restated: 2B x 2B matrix,
    diagonal = -inf, positives i <-> i + B, mean cross-entropy."""
    b = z0.shape[0]
    f = torch.cat([F.normalize(z0, dim=-1), F.normalize(z1, dim=-1)], 0)
    sim = f @ f.t() / temperature
    sim = sim.masked_fill(torch.eye(2 * b, dtype=torch.bool), float("-inf"))
    labels = torch.cat([torch.arange(b, 2 * b), torch.arange(0, b)])
    return F.cross_entropy(sim, labels)


def ntxent_f64(z0, z1, temperature):
    """float64 value + per-row log-sum-exp (what hcir_ntxent_fwd returns as row_lse)."""
    z0 = z0.double()
    z1 = z1.double()
    b = z0.shape[0]
    f = torch.cat([F.normalize(z0, dim=-1), F.normalize(z1, dim=-1)], 0)
    sim = f @ f.t() / temperature
    sim = sim.masked_fill(torch.eye(2 * b, dtype=torch.bool), float("-inf"))
    lse = torch.logsumexp(sim, dim=1)
    pos = sim[torch.arange(2 * b), torch.cat([torch.arange(b, 2 * b), torch.arange(0, b)])]
    return (lse - pos).mean(), lse
