"""world_size-8 gloo test of the sharded search (CPU): the control flow the driver's 8-GPU RCCL run executes first
 — eight ranks, UNEVEN shards, one shard with fewer rows than k (padding with empty slots),
one rank whose local search goes through a PendingSearch-style handle that falls back at finish() (the
certification path of hcir.gallery), packed single-collective exchanges.  The oracle stands in for the HIP ops;
every rank's result must equal a single scan of the whole gallery."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, K, D, NQ_LOCAL = 8, 10, 48, 3
BOUNDS = [0, 7, 140, 141, 400, 655, 656, 1200, 1711]   # rank 0 holds 7 < k rows, ranks 2 and 5 one row each


def _data():
    rng = np.random.default_rng(21)
    g = rng.standard_normal((BOUNDS[-1], D), dtype=np.float32)
    g[1500] = g[3]            # exact tie between the first (tiny) and the last shard
    g[140] = g[3]             # ... and a one-row shard
    q = rng.standard_normal((WORLD * NQ_LOCAL, D), dtype=np.float32)
    q[0] = g[3]
    return g, q


class OracleOps:
    @staticmethod
    def sim_topk(q, g, k, q_inv_norm=None, g_inv_norm=None, idx_base=0):
        from oracle import knn as oknn
        v, i = oknn.cosine_topk(q.numpy(), g.numpy(), k, idx_base=idx_base)
        return torch.from_numpy(v), torch.from_numpy(i)

    @staticmethod
    def topk_merge(vals, idx, k_out):
        from oracle import knn as oknn
        v, i = oknn.topk_merge(vals.numpy(), idx.numpy(), k_out)
        return torch.from_numpy(v), torch.from_numpy(i)


class _FallingBackResident:
    """Stands where hcir.gallery.ResidentGallery sits on a rank: search_begin returns a handle whose finish() first
    holds a WRONG filtered answer for two queries and repairs them through the exact scan, as the certified
    fallback does."""

    def __init__(self, shard, idx_base):
        self.shard, self.idx_base, self.fallbacks = shard, idx_base, 0

    def search_begin(self, q_all, k, q16=None):
        outer = self

        class Handle:
            def finish(self_h):
                val, idx = OracleOps.sim_topk(q_all, outer.shard, k, idx_base=outer.idx_base)
                val, idx = val.clone(), idx.clone()
                val[[1, 4]] = -7.0                      # "uncertified" rows of the filter pass
                bad = torch.tensor([1, 4])
                bv, bi = OracleOps.sim_topk(q_all[bad], outer.shard, k, idx_base=outer.idx_base)
                val[bad], idx[bad] = bv, bi
                outer.fallbacks += len(bad)
                return val, idx
        return Handle()


def _worker(rank, world, port, tmp):
    for p in (ROOT, os.path.join(ROOT, "hair-centric-image-retrieval_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hcir.dist import ShardedGallery
    g, q = _data()
    lo, hi = BOUNDS[rank], BOUNDS[rank + 1]
    shard = torch.from_numpy(g[lo:hi])
    resident = _FallingBackResident(shard, lo) if rank == 6 else None
    gal = ShardedGallery(shard, lo, ops=OracleOps, resident=resident)
    q_all = gal.gather_queries(torch.from_numpy(q[rank * NQ_LOCAL:(rank + 1) * NQ_LOCAL]))
    assert q_all.shape == (WORLD * NQ_LOCAL, D) and np.array_equal(q_all.numpy(), q)
    # two searches in flight, finished in order: the bench's software pipeline
    h1 = gal.search_begin(q_all, K)
    h2 = gal.search_begin(q_all[:5].contiguous(), 3)
    val, idx = gal.search_finish(h1)
    v2, i2 = gal.search_finish(h2)
    if resident is not None:
        assert resident.fallbacks == 4
    np.save(os.path.join(tmp, f"val{rank}.npy"), val.numpy())
    np.save(os.path.join(tmp, f"idx{rank}.npy"), idx.numpy())
    np.save(os.path.join(tmp, f"idx2_{rank}.npy"), i2.numpy())
    dist.destroy_process_group()


def test_sharded_search_eight_ranks_uneven(tmp_path):
    from oracle import knn as oknn
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    g, q = _data()
    rv, ri = oknn.cosine_topk(q, g, K)
    _, ri3 = oknn.cosine_topk(q[:5], g, 3)
    for r in range(WORLD):
        np.testing.assert_array_equal(np.load(tmp_path / f"idx{r}.npy"), ri)
        np.testing.assert_array_equal(np.load(tmp_path / f"val{r}.npy"), rv)
        np.testing.assert_array_equal(np.load(tmp_path / f"idx2_{r}.npy"), ri3)
    assert list(ri[0, :3]) == [3, 140, 1500]   # the tie spans three shards: smallest global index first
