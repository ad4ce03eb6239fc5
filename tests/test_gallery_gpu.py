"""GPU parity of the "exact by verification" search (hcir.gallery.ResidentGallery):
results must be BIT-IDENTICAL to the exact fp32 scan / the C oracle, whether a query is certified
from the fp16-mirror candidates or falls back to the full fp32 scan."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import knn as oknn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(hcir_built):
    assert torch.cuda.is_available()


def _unit(shape, seed):
    x = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("nq,ng,d,k", [(70, 60000, 768, 10), (5, 3000, 64, 5), (130, 40000, 256, 14), (3, 10, 8, 3)])
def test_filtered_search_is_exact(nq, ng, d, k):
    from hcir.gallery import ResidentGallery
    q, g = _unit((nq, d), 1), _unit((ng, d), 2)
    gal = ResidentGallery(torch.from_numpy(g).cuda(), idx_base=7)
    val, idx = gal.search(torch.from_numpy(q).cuda(), k)
    rv, ri = oknn.cosine_topk(q, g, k, idx_base=7)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)
    ev, ei = gal.search_exact(torch.from_numpy(q).cuda(), k)
    assert torch.equal(ei, idx) and torch.equal(ev, val)
    if ng > 16:
        assert gal.stats["queries"] == nq
        assert gal.stats["fallback_queries"] <= max(2, nq // 10)   # random data: almost all certified


def test_filtered_search_adversarial_near_duplicates():
    """Hundreds of gallery rows within 1e-5 of each other around the top of every query's ranking: the
    fp16 mirror cannot separate them, certification must fail and the fallback must restore exactness."""
    from hcir.gallery import ResidentGallery
    rng = np.random.default_rng(3)
    d, ng = 256, 20000
    g = _unit((ng, d), 4)
    q = _unit((12, d), 5)
    for i in range(6):                      # queries 0..5 get a cloud of 200 near-duplicate best matches
        base = q[i] + 0.05 * rng.standard_normal(d).astype(np.float32)
        rows = rng.choice(ng, 200, replace=False)
        g[rows] = base + 1e-5 * rng.standard_normal((200, d)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g[4321] = g[77]                          # exact duplicate rows: tie -> smaller index
    gal = ResidentGallery(torch.from_numpy(g).cuda())
    val, idx = gal.search(torch.from_numpy(q).cuda(), 10)
    rv, ri = oknn.cosine_topk(q, g, 10)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)
    assert gal.stats["fallback_queries"] >= 6     # the clouds cannot be certified
    assert gal.stats["fallback_queries"] < 12     # the ordinary queries are


def test_error_bound_holds():
    """The bound E_i used for certification really bounds |exact fp32 chain - fp16-mirror score|."""
    from hcir import ops
    from hcir.gallery import ResidentGallery
    q, g = _unit((16, 768), 6), _unit((5000, 768), 7)
    gal = ResidentGallery(torch.from_numpy(g).cuda())
    qd = torch.from_numpy(q).cuda()
    q16 = qd.half()
    err = gal.err_bound(qd, q16).cpu().numpy()
    s16, i16 = ops.sim_topk(q16, gal.mirror, 16)
    exact = oknn.scores(q, g)                                   # fp32 chain, all pairs
    picked = np.take_along_axis(exact, i16.cpu().numpy(), axis=1)
    diff = np.abs(picked - s16.cpu().numpy())
    assert (diff <= err[:, None]).all()
    assert err.max() < 2e-3 and diff.max() < err.max()          # and it is not vacuous


def test_in_kernel_bound_matches_python_bound():
    """hcir_topk_refine_f32 derives E_i itself from the mirror constants; it must certify exactly the
    queries the documented (python) bound certifies."""
    from hcir import _lib, ops
    from hcir.gallery import FILTER_KC, ResidentGallery
    q, g = _unit((64, 256), 10), _unit((20000, 256), 11)
    g[100:400] = g[50] + 2e-4 * np.random.default_rng(0).standard_normal((300, 256)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    q[:8] = g[50] + 0.02 * np.random.default_rng(1).standard_normal((8, 256)).astype(np.float32)
    gal = ResidentGallery(torch.from_numpy(g).cuda())
    qd = torch.from_numpy(q).cuda()
    q16 = qd.half()
    cval, cidx = ops.sim_topk(q16, gal.mirror, FILTER_KC)
    outs = []
    for err in (gal.err_bound(qd, q16), None):
        val = torch.empty((64, 10), device="cuda")
        idx = torch.empty((64, 10), dtype=torch.int64, device="cuda")
        cert = torch.empty(64, dtype=torch.int32, device="cuda")
        assert _lib.lib().hcir_topk_refine_f32(
            qd.data_ptr(), 64, gal.g32.data_ptr(), 20000, 256, cidx.data_ptr(), cval.data_ptr(), FILTER_KC, 10, 0,
            None, None, None if err is None else err.data_ptr(), None if err is not None else gal._consts,
            val.data_ptr(), idx.data_ptr(), cert.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        outs.append((val.cpu(), idx.cpu(), cert.cpu()))
    assert torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][1], outs[1][1])
    assert 0 < int((outs[1][2] == 0).sum()) < 64      # the planted cloud defeats some queries, not all


def test_sharded_gallery_uses_resident():
    from hcir.dist import ShardedGallery
    from hcir.gallery import ResidentGallery
    q, g = _unit((33, 128), 8), _unit((30000, 128), 9)
    gd = torch.from_numpy(g).cuda()
    sg = ShardedGallery(gd, 100, resident=ResidentGallery(gd, 100))
    val, idx = sg.search(torch.from_numpy(q).cuda(), 10)
    rv, ri = oknn.cosine_topk(q, g, 10, idx_base=100)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(val.cpu().numpy(), rv)
