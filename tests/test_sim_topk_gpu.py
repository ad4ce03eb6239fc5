"""GPU parity: hcir_sim_topk / hcir_topk_merge / hcir_row_invnorm vs oracle/knn_oracle.c.

HCIR_F32 is compared BIT-EXACTLY (values and indices): libhcir's score is one fp32
fmaf chain in a documented k-order which the C oracle follows.  fp16/bf16 storage is
compared against the float64 oracle within a stated tolerance, indices exact wherever
the oracle's score gaps exceed that tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import knn as oknn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(hcir_built):
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from hcir import ops as _ops
    return _ops


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


CASES_F32 = [
    # nq, ng, d, k
    (1, 37, 8, 5),          # single query, tiny ragged gallery, minimum d
    (5, 1000, 64, 10),
    (64, 1000, 2048, 5),    # BASELINE config C1 shape
    (64, 10000, 768, 10),   # BASELINE config C2 shape
    (33, 777, 72, 16),      # ragged everything, d not a multiple of the 32-chunk
    (128, 4099, 768, 10),
    (200, 3000, 128, 7),    # two query blocks
    (16, 50000, 768, 10),   # two-phase (prefix + floor) path
    (70, 40000, 256, 3),
]


@pytest.mark.parametrize("nq,ng,d,k", CASES_F32)
def test_sim_topk_f32_bit_exact(ops, nq, ng, d, k):
    q, g = _rand((nq, d), 1), _rand((ng, d), 2)
    val, idx = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), k)
    rv, ri = oknn.cosine_topk(q, g, k)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)


def test_row_invnorm_bit_exact(ops):
    for n, d in ((7, 8), (1000, 768), (33, 2048), (5, 1000)):
        x = _rand((n, d), 3) * 3.0
        out = ops.row_invnorm(torch.from_numpy(x).cuda(), 1e-12)
        np.testing.assert_array_equal(out.cpu().numpy(), oknn.row_invnorm(x, 1e-12))


def test_sim_topk_f32_norms_bit_exact(ops):
    """cosine of un-normalised rows: (<g,q> * gn) * qn, the hair_encoder.py:193 path."""
    q, g = _rand((9, 768), 4) * 2.5, _rand((5000, 768), 5) * 0.3
    qd, gd = torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda()
    qn, gn = ops.row_invnorm(qd, 1e-12), ops.row_invnorm(gd, 1e-12)
    val, idx = ops.sim_topk(qd, gd, 5, q_inv_norm=qn, g_inv_norm=gn, idx_base=1000)
    rv, ri = oknn.cosine_topk(q, g, 5, qn=oknn.row_invnorm(q, 1e-12), gn=oknn.row_invnorm(g, 1e-12),
                              idx_base=1000)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)


def test_sim_topk_planted_ties(ops):
    """Exact ties resolve to the smaller index (documented tie-break)."""
    g = _rand((3000, 64), 6)
    g[17] = g[5]
    g[2999] = g[5]
    g[1234] = g[5]
    q = np.stack([g[5] * 2.0, g[100]]).astype(np.float32)
    val, idx = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), 6)
    rv, ri = oknn.cosine_topk(q, g, 6)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    assert list(idx[0, :4].cpu().numpy()) == [5, 17, 1234, 2999]


@pytest.mark.parametrize("k", [17, 64, 100, 642])
def test_sim_topk_large_k(ops, k):
    """k up to 642 is in the reference's sweep (HP/src/classification_engine.py:71)."""
    q, g = _rand((20, 128), 7), _rand((3000, 128), 8)
    val, idx = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), k)
    rv, ri = oknn.cosine_topk(q, g, k)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)


def test_sim_topk_k_equals_ng_and_errors(ops):
    q, g = _rand((3, 16), 9), _rand((12, 16), 10)
    val, idx = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), 12)
    rv, ri = oknn.cosine_topk(q, g, 12)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    with pytest.raises(ValueError):  # sklearn raises when n_neighbors > n_samples_fit
        ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), 13)
    from hcir import HcirError
    with pytest.raises(HcirError):  # no CPU fallback
        ops.sim_topk(torch.from_numpy(q), torch.from_numpy(g), 3)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("nq,ng,d,k", [(64, 10000, 768, 10), (128, 60000, 768, 10), (3, 500, 64, 5)])
def test_sim_topk_reduced_precision(ops, dtype, tol, nq, ng, d, k):
    """fp16/bf16 storage: compare against float64 scores of the SAME rounded inputs.
    Tolerance `tol` is absolute on unit-norm cosine scores (fp32 accumulate of exact
    products: the only error is fp32 summation, so it is far tighter than input rounding)."""
    q, g = _rand((nq, d), 11), _rand((ng, d), 12)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    qd, gd = torch.from_numpy(q).cuda().to(dtype), torch.from_numpy(g).cuda().to(dtype)
    val, idx = ops.sim_topk(qd, gd, k)
    qr, gr = qd.float().cpu().numpy(), gd.float().cpu().numpy()   # rounded inputs, exact in fp32
    s = qr.astype(np.float64) @ gr.astype(np.float64).T
    rv, ri = oknn.stable_topk_np(s, k + 1)
    val, idx = val.cpu().numpy(), idx.cpu().numpy()
    np.testing.assert_allclose(val, rv[:, :k], atol=1e-5, rtol=0)
    gap = rv[:, :-1] - rv[:, 1:]            # gaps between consecutive oracle ranks
    safe = np.minimum(np.concatenate([np.full((nq, 1), np.inf), gap[:, :-1]], 1), gap) > 1e-5
    np.testing.assert_array_equal(idx[safe], ri[:, :k][safe])
    # and against the unrounded fp32 inputs within the storage tolerance
    s32 = q.astype(np.float64) @ g.astype(np.float64).T
    rv32, _ = oknn.stable_topk_np(s32, k)
    np.testing.assert_allclose(val, rv32, atol=tol, rtol=0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nq,ng,d,k", [(129, 70000, 768, 10), (220, 120000, 768, 16), (300, 50000, 128, 16),
                                       (600, 40000, 64, 1)])
def test_sim_topk_many_queries_big_tile(ops, dtype, nq, ng, d, k):
    """nq > 128, k <= 16, fp16/bf16: the 256 x 256 tile scan (candidates above the prefix floor through an
    atomic append, no register lists).  Scores are bit-identical to the list-keeping kernel (same MFMA chain),
    which the same gallery reaches with <= 128 queries: compare block by block, values AND indices exactly."""
    q, g = _rand((nq, d), 21), _rand((ng, d), 22)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    qd, gd = torch.from_numpy(q).cuda().to(dtype), torch.from_numpy(g).cuda().to(dtype)
    val, idx = ops.sim_topk(qd, gd, k, idx_base=7)
    for s0 in range(0, nq, 128):          # <= 128 queries: sim_topk_scan with register lists
        rv, ri = ops.sim_topk(qd[s0:s0 + 128].contiguous(), gd, k, idx_base=7)
        np.testing.assert_array_equal(idx[s0:s0 + 128].cpu().numpy(), ri.cpu().numpy())
        np.testing.assert_array_equal(val[s0:s0 + 128].cpu().numpy(), rv.cpu().numpy())
    # and against float64 scores of the rounded inputs
    s = qd.float().cpu().numpy().astype(np.float64) @ gd.float().cpu().numpy().astype(np.float64).T
    rv64, _ = oknn.stable_topk_np(s, k)
    np.testing.assert_allclose(val.cpu().numpy(), rv64, atol=1e-5, rtol=0)


def test_sim_topk_big_tile_overflow_falls_back(ops):
    """A gallery whose later rows systematically beat its first ones: every row behind the prefix is a
    near-duplicate of the queries' mean direction, so far more than 512 rows per query clear the prefix floor,
    the candidate buffers overflow and the device-side gate runs the list-keeping scan.  Same exact result."""
    rng = np.random.default_rng(31)
    nq, ng, d, k = 200, 60000, 128, 16
    base = rng.standard_normal(d).astype(np.float32)
    q = (base[None] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    g = rng.standard_normal((ng, d)).astype(np.float32)                     # prefix: unrelated rows
    g[8192:] = base[None] + 0.3 * rng.standard_normal((ng - 8192, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    qd, gd = torch.from_numpy(q).cuda().half(), torch.from_numpy(g).cuda().half()
    val, idx = ops.sim_topk(qd, gd, k)
    for s0 in range(0, nq, 100):
        rv, ri = ops.sim_topk(qd[s0:s0 + 100].contiguous(), gd, k)
        np.testing.assert_array_equal(idx[s0:s0 + 100].cpu().numpy(), ri.cpu().numpy())
        np.testing.assert_array_equal(val[s0:s0 + 100].cpu().numpy(), rv.cpu().numpy())
    assert (idx.cpu().numpy() >= 8192).all()      # the winners really are behind the prefix


def test_topk_merge(ops):
    rng = np.random.default_rng(13)
    nl, nq, kin, kout = 8, 50, 10, 10
    vals = np.sort(rng.standard_normal((nl, nq, kin)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    idx = np.empty((nl, nq, kin), dtype=np.int64)
    for l in range(nl):
        idx[l] = l * 1000 + np.sort(rng.integers(0, 1000, (nq, kin)), axis=1)
    vals[3, :, 7:] = -np.inf   # ragged shard: empty slots
    idx[3, :, 7:] = -1
    vals[1, 0, 0] = vals[2, 0, 0] = 5.0  # cross-shard tie -> smaller index
    ov, oi = ops.topk_merge(torch.from_numpy(vals).cuda(), torch.from_numpy(idx).cuda(), kout)
    rv, ri = oknn.topk_merge(vals, idx, kout)
    np.testing.assert_array_equal(oi.cpu().numpy(), ri)
    np.testing.assert_array_equal(ov.cpu().numpy(), rv)


# ---- config C5 geometry (BASELINE.json configs[4]): d = 1024, top-50, fp16 / bf16 storage -------------------
def _scores64_rounded(qd, gd, chunk=65536):
    """float64 scores of the ROUNDED inputs (exact products, exact sums up to fp64), gallery in chunks."""
    qr = qd.float().cpu().numpy().astype(np.float64)
    out = np.empty((qr.shape[0], gd.shape[0]), dtype=np.float64)
    for s0 in range(0, gd.shape[0], chunk):
        out[:, s0:s0 + chunk] = qr @ gd[s0:s0 + chunk].float().cpu().numpy().astype(np.float64).T
    return out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nq,ng", [(32, 40000), (64, 40000), (200, 40000), (32, 300000), (64, 300000),
                                   (200, 300000)])
def test_sim_topk_c5_top50_d1024(ops, dtype, nq, ng):
    """The 64-entry-list kernel (16 < k <= 64) with its 4-slot DMA ring, with and without the prefix pass
    (N >= 32768 takes it; the 300 000-row gallery runs a 16 K-row prefix), fp16 AND bf16, at config C5's row
    length and k.  Checked against float64 scores of the same rounded inputs: values within
    1e-5, indices exact wherever the oracle's neighbouring gaps exceed that."""
    d, k = 1024, 50
    q, g = _rand((nq, d), 41), _rand((ng, d), 42)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g[ng - 5] = g[11]                        # an exact duplicate far behind the prefix
    q[0] = g[11]
    qd, gd = torch.from_numpy(q).cuda().to(dtype), torch.from_numpy(g).cuda().to(dtype)
    val, idx = ops.sim_topk(qd, gd, k, idx_base=3)
    val, idx = val.cpu().numpy(), idx.cpu().numpy() - 3
    rv, ri = oknn.stable_topk_np(_scores64_rounded(qd, gd), k + 1)
    np.testing.assert_allclose(val, rv[:, :k], atol=1e-5, rtol=0)
    gap = rv[:, :-1] - rv[:, 1:]
    safe = np.minimum(np.concatenate([np.full((nq, 1), np.inf), gap[:, :-1]], 1), gap) > 1e-5
    np.testing.assert_array_equal(idx[safe], ri[:, :k][safe])
    assert safe.mean() > 0.9                 # the comparison is not vacuous
    assert list(idx[0, :2]) == [11, ng - 5]  # identical rows give identical fp32 chains: tie -> smaller index
    assert (np.diff(val, axis=1) <= 0).all()


@pytest.mark.parametrize("nq,ng,d,k", [(32, 40000, 1024, 50), (64, 50000, 768, 50), (130, 70000, 256, 33),
                                       (16, 300000, 128, 64)])
def test_sim_topk_f32_top50_bit_exact(ops, nq, ng, d, k):
    """fp32 storage through the same 64-entry-list + prefix path: bit-exact against the C oracle."""
    q, g = _rand((nq, d), 43), _rand((ng, d), 44)
    g[ng - 1] = g[2]
    q[1] = g[2]
    val, idx = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), k)
    rv, ri = oknn.cosine_topk(q, g, k)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("nq,k", [(64, 10), (64, 50), (20, 16), (128, 33)])
def test_sim_topk_candidate_overflow_falls_back(ops, dtype, nq, k):
    """<= 128 queries run the candidate-append flow (lists only in the prefix, every later row at / above the
    floor appended to a per-query buffer).  A gallery ordered so that the later rows systematically beat the
    first ones (a class-ordered gallery and queries of a late class) overflows the buffers: the device flag must
    route the call through the list-keeping flow, same exact result."""
    rng = np.random.default_rng(51)
    ng, d = 60000, 128
    base = rng.standard_normal(d).astype(np.float32)
    q = (base[None] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    g = rng.standard_normal((ng, d)).astype(np.float32)
    g[16384:] = base[None] + 0.3 * rng.standard_normal((ng - 16384, d)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    qd, gd = torch.from_numpy(q).cuda().to(dtype), torch.from_numpy(g).cuda().to(dtype)
    val, idx = ops.sim_topk(qd, gd, k, idx_base=9)
    val, idx = val.cpu().numpy(), idx.cpu().numpy() - 9
    assert (idx >= 16384).all()
    if dtype == torch.float32:
        rv, ri = oknn.cosine_topk(q, g, k)
        np.testing.assert_array_equal(idx, ri)
        np.testing.assert_array_equal(val, rv)
    else:
        rv, ri = oknn.stable_topk_np(_scores64_rounded(qd, gd), k + 1)
        np.testing.assert_allclose(val, rv[:, :k], atol=1e-5, rtol=0)
        gap = rv[:, :-1] - rv[:, 1:]
        safe = np.minimum(np.concatenate([np.full((nq, 1), np.inf), gap[:, :-1]], 1), gap) > 1e-5
        np.testing.assert_array_equal(idx[safe], ri[:, :k][safe])


@pytest.mark.parametrize("k", [10, 16, 17, 50, 64])
def test_sim_topk_candidate_flow_ties_across_prefix(ops, k):
    """Exact duplicates of the best rows on both sides of the prefix boundary and inside the prefix: the
    candidate flow's `> floor` (k <= 16) / `>= floor` (group floors, k > 16) rules must keep the tie-break
    (smaller index first) bit-exactly, fp32."""
    rng = np.random.default_rng(52)
    nq, ng, d = 40, 70000, 64
    q, g = rng.standard_normal((nq, d), dtype=np.float32), rng.standard_normal((ng, d), dtype=np.float32)
    for i in range(nq):                       # every query's best row is repeated k + 3 times over the gallery
        best = g[100 + i].copy()
        q[i] = best
        pos = rng.choice(np.arange(200, ng), size=k + 2, replace=False)
        g[pos] = best
    val, idx = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(g).cuda(), k)
    rv, ri = oknn.cosine_topk(q, g, k)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)
