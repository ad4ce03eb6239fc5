"""hcir.neg_sampling — NegSamplerStatic with the reference's signature
(HP/src/neg_sampling.py:26-53): B x B cosine of the momentum-encoder embeddings,
rank-k pick per row.  The reference sorts every row fully (torch.sort) and takes
column k-1; here hcir_sim_topk selects the k best and column k-1 is returned
(self-similarity is rank 0, as in the reference).
"""
from __future__ import annotations

import torch

from . import ops


def NegSamplerStatic(model, batch, metric="cosine", k=7):
    with torch.no_grad():
        embeddings = model.extract_features_ema(batch)
    B, D = embeddings.shape
    if metric == "cosine":
        # embeddings / norm.clamp(min=1e-8): folded into the scan as per-row factors
        emb = embeddings.float().contiguous()
        inv = ops.row_invnorm(emb, 1e-8)
    elif metric == "euclidean":
        raise NotImplementedError("only the cosine metric is on the HIP path "
                                  "(the reference's scripts never pass 'euclidean')")
    else:
        raise ValueError("Unsupported metric. Choose 'cosine' or 'euclidean'.")
    if k < 1 or k > B:
        raise ValueError(f"k must be between 1 and {B}")
    _, idx = ops.sim_topk(emb, emb, k, q_inv_norm=inv, g_inv_norm=inv)
    return idx[:, k - 1]


def NegSamplerRandomly(embeddings: torch.Tensor):
    """HP/src/neg_sampling.py:10-23, vectorised: random permutation, fixed points shifted by one."""
    batch_size = embeddings.size(0)
    perm = torch.randperm(batch_size, device=embeddings.device)
    ar = torch.arange(batch_size, device=embeddings.device)
    perm = torch.where(perm == ar, (perm + 1) % batch_size, perm)
    return embeddings[perm]
