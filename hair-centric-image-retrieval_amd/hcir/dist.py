"""hcir.dist — gallery row-sharding across the GPUs of one node.

The reference is single-GPU with the kNN on the CPU: there is no NCCL
call pattern to translate.  The sharded search is new capability defined by north_star:

  rank r holds gallery rows [r*N/P, (r+1)*N/P)   (contiguous, idx_base = first row)
  1. all-gather of the query embeddings   [Q/P, D] per rank   (RCCL over xGMI)
  2. local hcir_sim_topk of ALL Q queries against the local shard -> (val, idx)[Q, k]
     with GLOBAL row indices
  3. all-gather of the per-shard top-k    Q*k*12 B per rank, ONE collective: (fp32 value, int64 index) travel
     as one int32 record [Q, 3k] (value bits | index words) into one preallocated [P, Q, 3k] buffer
  4. hcir_topk_merge of the P lists with the global tie-break (score desc, index asc)
     -> identical to a single-GPU scan of the whole gallery.

Both payloads are latency-bound (KBs); one process per GPU, torch.distributed
(backend "nccl" == RCCL on ROCm).  `ops` is injectable so the world_size-2 gloo tests
can drive the same control flow on CPU with the oracle as the checker's stand-in.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous near-equal split: the first n_rows % world ranks get one extra row."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_into(tensor: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """ONE all-gather of equal-shape tensors into a [world, *shape] buffer (no list of outputs, no stack / cat
    afterwards).  RCCL ("nccl") gathers device tensors in place; under the gloo REHEARSAL backend (several
    ranks sharing one GPU, CPU tests) the payload hops through the host."""
    tensor = tensor.contiguous()
    if dist.get_backend(group) == "gloo" and tensor.is_cuda:
        host = tensor.cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)   # (concatenated along dim 0: the form every backend takes)
        return out.to(tensor.device).view((world,) + tuple(tensor.shape))
    out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    dist.all_gather_into_tensor(out, tensor, group=group)
    return out.view((world,) + tuple(tensor.shape))


def pack_topk(val: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """(fp32 [Q,k], int64 [Q,k]) -> int32 [Q, 3k]: the value bits, then the index as two 32-bit words."""
    q, k = val.shape
    rec = torch.empty((q, 3 * k), dtype=torch.int32, device=val.device)
    rec[:, :k] = val.contiguous().view(torch.int32)
    rec[:, k:] = idx.contiguous().view(torch.int32)
    return rec


def unpack_topk(rec: torch.Tensor, k: int):
    """int32 [..., 3k] -> (fp32 [..., k], int64 [..., k])."""
    val = rec[..., :k].contiguous().view(torch.float32)
    idx = rec[..., k:].contiguous().view(torch.int64)
    return val, idx


class ShardedGallery:
    """One rank's shard of an L2-normalised gallery plus the collective search."""

    def __init__(self, shard: torch.Tensor, idx_base: int, group=None, ops=None,
                 g_inv_norm: Optional[torch.Tensor] = None, resident=None):
        if ops is None:
            from . import ops as _ops  # HIP path; raises if libhcir.so is missing
            ops = _ops
        self.ops = ops
        self.shard = shard
        self.idx_base = int(idx_base)
        self.group = group
        self.g_inv_norm = g_inv_norm
        # optional hcir.gallery.ResidentGallery of the SAME shard: exact fp32 results through the
        # fp16-mirror filter + exact refine + certified fallback
        self.resident = resident
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def gather_queries(self, q_local: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return q_local
        out = _all_gather_into(q_local, self.world, self.group)      # [P, Q/P, D], rank-major == row order
        return out.view(-1, q_local.shape[1])

    def search_begin(self, q_all: torch.Tensor, k: int, q_inv_norm: Optional[torch.Tensor] = None,
                     q16: Optional[torch.Tensor] = None):
        """Enqueue the local scan of this rank's shard; returns a handle for search_finish.  `q16`: the fp16 copy of
        q_all when the caller already has it (the engine's normalise kernel writes both)."""
        k_local = min(k, self.shard.shape[0])
        if self.resident is not None and q_inv_norm is None and self.g_inv_norm is None:
            return (self.resident.search_begin(q_all, k_local, q16=q16), k, k_local)
        val, idx = self.ops.sim_topk(q_all, self.shard, k_local, q_inv_norm=q_inv_norm,
                                     g_inv_norm=self.g_inv_norm, idx_base=self.idx_base)
        return ((val, idx), k, k_local)

    def search_finish(self, handle):
        """Certify / fall back locally, then all-gather the per-shard top-k and merge (same result on
        every rank, identical to a single scan of the whole gallery)."""
        local, k, k_local = handle
        val, idx = local.finish() if hasattr(local, "finish") else local
        if k_local < k:  # tiny shard: pad with empty slots
            pad_v = torch.full((val.shape[0], k - k_local), float("-inf"), dtype=val.dtype, device=val.device)
            pad_i = torch.full((idx.shape[0], k - k_local), -1, dtype=idx.dtype, device=idx.device)
            val, idx = torch.cat([val, pad_v], 1), torch.cat([idx, pad_i], 1)
        if self.world == 1:
            return val, idx
        rec = _all_gather_into(pack_topk(val, idx), self.world, self.group)   # [P, Q, 3k]
        vals, idxs = unpack_topk(rec, k)
        return self.ops.topk_merge(vals, idxs, k)

    def search(self, q_all: torch.Tensor, k: int, q_inv_norm: Optional[torch.Tensor] = None):
        """Top-k of every query in q_all over the WHOLE (sharded) gallery; same result on all ranks."""
        return self.search_finish(self.search_begin(q_all, k, q_inv_norm))
