"""Full-size parity of config C4 (BASELINE.json configs[3]): 1 000 000 x 768 gallery, top-10.

VERDICT r1 weak #3: the bench's own geometry (prefix ~ N/16, 880 queries per rank, candidate buffers; 7 040
queries against 125 000-row shards at 8 GPUs) had never been compared with the oracle.  The C oracle's
blocked scan (bit-identical to its scalar chain, tests/test_oracle_pins.py) makes that a seconds-long check.

  * hcir_sim_topk(HCIR_F32) and ResidentGallery.search (fp16-mirror filter + exact refine + certified
    fallback) against oknn.cosine_topk, values AND indices with assert_array_equal, 64 and 880 queries;
  * the 8-way sharded search in ONE process: eight 125 000-row shards with global index bases, each searched
    with all 8 x 880 = 7 040 queries exactly as rank r would, then hcir_topk_merge — the P = 8 geometry the
    driver's RCCL run uses.  (Eight GPU processes on one card would exceed the box's process guard of 6; the
    collectives themselves are covered by tests/test_dist_gloo.py and tests/test_bench_multirank_gpu.py.)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import knn as oknn

pytestmark = pytest.mark.gpu

N, D, K = 1_000_000, 768, 10


@pytest.fixture(scope="module")
def big(hcir_built):
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    g = torch.empty((N, D), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1000)
    for s in range(0, N, 131072):
        e = min(N, s + 131072)
        g[s:e] = F.normalize(torch.randn((e - s, D), generator=gen, device=dev), dim=1)
    g[N - 7] = g[123]                              # exact duplicates: one far behind every prefix / shard
    g[500_000] = g[123]
    q = F.normalize(torch.randn((7040, D), generator=torch.Generator(device=dev).manual_seed(1), device=dev), dim=1)
    q[0] = g[123]
    q[5000] = g[123]
    g_np, q_np = g.cpu().numpy(), q.cpu().numpy()
    rv, ri = oknn.cosine_topk(q_np, g_np, K)       # the whole 7 040 x 1 M oracle, once
    assert list(ri[0, :3]) == [123, 500_000, N - 7]
    return dict(g=g, q=q, rv=rv, ri=ri)


@pytest.mark.parametrize("nq", [64, 880])
def test_c4_fullsize_exact_scan(big, nq):
    from hcir import ops
    val, idx = ops.sim_topk(big["q"][:nq].contiguous(), big["g"], K)
    np.testing.assert_array_equal(idx.cpu().numpy(), big["ri"][:nq])
    np.testing.assert_array_equal(val.cpu().numpy(), big["rv"][:nq])


@pytest.mark.parametrize("nq", [64, 880])
def test_c4_fullsize_filtered_search(big, nq):
    from hcir.gallery import ResidentGallery
    gal = ResidentGallery(big["g"])
    val, idx = gal.search(big["q"][:nq].contiguous(), K)
    np.testing.assert_array_equal(idx.cpu().numpy(), big["ri"][:nq])
    np.testing.assert_array_equal(val.cpu().numpy(), big["rv"][:nq])
    assert gal.stats["queries"] == nq
    # the planted exact duplicates tie in the mirror too; everything else certifies
    assert gal.stats["fallback_queries"] <= max(2, nq // 50)


def test_c4_eight_way_sharded_geometry(big):
    """Every 'rank' searches ALL 7 040 queries against its 125 000-row shard (global indices), the eight
    per-shard top-10 lists are merged with the global tie-break: identical to the single scan."""
    from hcir import ops
    from hcir.dist import shard_bounds
    from hcir.gallery import ResidentGallery
    vals, idxs = [], []
    for r in range(8):
        lo, hi = shard_bounds(N, 8, r)
        assert hi - lo == 125_000
        gal = ResidentGallery(big["g"][lo:hi], lo)
        v, i = gal.search(big["q"], K)
        vals.append(v)
        idxs.append(i)
        assert int(i.min()) >= lo and int(i.max()) < hi
        del gal
    val, idx = ops.topk_merge(torch.stack(vals, 0), torch.stack(idxs, 0), K)
    np.testing.assert_array_equal(idx.cpu().numpy(), big["ri"])
    np.testing.assert_array_equal(val.cpu().numpy(), big["rv"])
    # the f16-only scan of one shard at the same shape (the kernel the timed path streams)
    lo, hi = shard_bounds(N, 8, 3)
    q16, g16 = big["q"].half(), big["g"][lo:hi].half()
    v16, i16 = ops.sim_topk(q16, g16, 16, idx_base=lo)
    for s0 in (0, 3456, 6912):                     # against the list-keeping kernel (<= 128 queries), exactly
        rv, ri = ops.sim_topk(q16[s0:s0 + 128].contiguous(), g16, 16, idx_base=lo)
        assert torch.equal(i16[s0:s0 + 128], ri) and torch.equal(v16[s0:s0 + 128], rv)


@pytest.mark.parametrize("nq", [32, 64])
def test_c5_full_shard_top50(hcir_built, nq):
    """Config C5's per-GPU shard at its REAL size (BASELINE.json configs[4]: 10 M x 1024 fp16 over 8 GPUs =
    1 250 000 rows each), k = 50: the candidate flow's four group floors and its 3008-entry buffers are sized from
    N, and VERDICT r2 weak #3 found them timed at this N but checked only up to 300 000 rows.  Reference: float64
    scores of the SAME rounded fp16 inputs (products of two fp16 are exact in fp64; computed on the device in row
    chunks), stable top-51.  Values within 1e-5; indices exact wherever the neighbouring reference gaps exceed that."""
    from hcir import ops
    n, d, k = 1_250_000, 1024, 50
    dev = torch.device("cuda", 0)
    g = torch.empty((n, d), dtype=torch.float16, device=dev)
    gen = torch.Generator(device=dev).manual_seed(2000)
    for s in range(0, n, 131072):
        e = min(n, s + 131072)
        g[s:e] = F.normalize(torch.randn((e - s, d), generator=gen, device=dev), dim=1).half()
    q = F.normalize(torch.randn((nq, d), generator=torch.Generator(device=dev).manual_seed(3), device=dev), dim=1).half()
    g[n - 9] = g[77]                       # exact duplicates: first rows vs the very end of the shard
    q[1] = g[77]
    val, idx = ops.sim_topk(q, g, k, idx_base=5)
    val, idx = val.cpu().numpy(), idx.cpu().numpy() - 5
    # float64 reference, running top-(k+1) over row chunks (score desc, index asc via stable sort)
    q64 = q.double()
    best_v = torch.full((nq, 0), 0.0, dtype=torch.float64, device=dev)
    best_i = torch.zeros((nq, 0), dtype=torch.int64, device=dev)
    for s in range(0, n, 125_000):
        e = min(n, s + 125_000)
        sc = q64 @ g[s:e].double().t()
        cv = torch.cat([best_v, sc], 1)
        ci = torch.cat([best_i, torch.arange(s, e, device=dev).expand(nq, -1)], 1)
        order = torch.sort(-cv, dim=1, stable=True).indices[:, :k + 1]   # candidates are in index order: ties keep it
        best_v, best_i = torch.gather(cv, 1, order), torch.gather(ci, 1, order)
    rv, ri = best_v.cpu().numpy(), best_i.cpu().numpy()
    np.testing.assert_allclose(val, rv[:, :k], atol=1e-5, rtol=0)
    gap = rv[:, :-1] - rv[:, 1:]
    safe = np.minimum(np.concatenate([np.full((nq, 1), np.inf), gap[:, :-1]], 1), gap) > 1e-5
    np.testing.assert_array_equal(idx[safe], ri[:, :k][safe])
    assert safe.mean() > 0.9
    assert list(idx[1, :2]) == [77, n - 9]
    assert (np.diff(val, axis=1) <= 0).all()
