"""GPU parity of the device-side class vote and retrieval metrics:
hcir_knn_vote vs oracle.knn_vote and live sklearn .predict for the reference's whole k sweep,
hcir_confusion_matrix vs sklearn, hcir_retrieval_metrics vs the restated quantitative_eval loop."""
import numpy as np
import pytest
import torch

from oracle import knn as oknn
from oracle import metrics as ometrics

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu(hcir_built):
    assert torch.cuda.is_available()


@pytest.mark.parametrize("nq,ntrain,nclass,ks", [
    (300, 5000, 31, (5, 10, 20, 27, 30, 40, 642)),     # the reference's sweep (HP/src/classification_engine.py:71)
    (7, 50, 3, (1, 2, 50)),
    (65, 3000, 1000, (3, 64, 65, 200)),                # more classes than lanes, segment edges at the wave width
])
def test_knn_vote_vs_oracle(nq, ntrain, nclass, ks):
    from hcir import metrics
    rng = np.random.default_rng(0)
    labels = rng.integers(0, nclass, ntrain)
    nbr = np.stack([rng.permutation(ntrain)[: ks[-1]] for _ in range(nq)]).astype(np.int64)
    pred = metrics.knn_vote(torch.from_numpy(nbr + 11).cuda(), torch.from_numpy(labels).cuda(), ks, nclass,
                            idx_base=11).cpu().numpy()
    for j, k in enumerate(ks):
        np.testing.assert_array_equal(pred[j], oknn.knn_vote(nbr[:, :k].copy(), labels, nclass))
    # planted tie: labels 2,1,1,2 -> 1
    lab = torch.tensor([2, 1, 1, 2, 0], device="cuda")
    p = metrics.knn_vote(torch.tensor([[0, 1, 2, 3]], device="cuda"), lab, (1, 4), 3).cpu().numpy()
    assert p[0, 0] == 2 and p[1, 0] == 1
    with pytest.raises(ValueError):
        metrics.knn_vote(torch.tensor([[0, -1]], device="cuda"), lab, (2,), 3)      # empty slot
    with pytest.raises(ValueError):
        metrics.knn_vote(torch.tensor([[0, 1]], device="cuda"), lab, (2, 1), 3)     # ks not ascending


def test_knn_vote_and_confusion_vs_live_sklearn():
    """End to end against the library holding the reference's arithmetic: top-max(k) from hcir_sim_topk,
    one vote launch for the sweep, vs KNeighborsClassifier(n_neighbors=k).fit().predict() per k."""
    from sklearn.metrics import accuracy_score, confusion_matrix
    from sklearn.neighbors import KNeighborsClassifier
    from hcir import metrics, ops
    rng = np.random.default_rng(1)
    centers = rng.standard_normal((12, 64)).astype(np.float32)
    ytr = rng.integers(0, 12, 2000)
    yte = rng.integers(0, 12, 150)
    xtr = (centers[ytr] + 1.5 * rng.standard_normal((2000, 64))).astype(np.float32)
    xte = (centers[yte] + 1.5 * rng.standard_normal((150, 64))).astype(np.float32)
    q, g = torch.from_numpy(xte).cuda(), torch.from_numpy(xtr).cuda()
    ks = (5, 10, 20, 27, 30, 40, 642)
    _, idx = ops.sim_topk(q, g, ks[-1], q_inv_norm=ops.row_invnorm(q, 1e-30), g_inv_norm=ops.row_invnorm(g, 1e-30))
    pred = metrics.knn_vote(idx, torch.from_numpy(ytr).cuda(), ks, 12)
    for j, k in enumerate(ks):
        ref = KNeighborsClassifier(n_neighbors=k, metric="cosine").fit(xtr, ytr).predict(xte)
        np.testing.assert_array_equal(pred[j].cpu().numpy(), ref)
        cm = metrics.confusion_matrix(torch.from_numpy(yte).cuda(), pred[j].contiguous(), 12).cpu().numpy()
        np.testing.assert_array_equal(cm, confusion_matrix(yte, ref, labels=np.arange(12)))
        assert cm.trace() / 150 == accuracy_score(yte, ref)


@pytest.mark.parametrize("nq,ndb,gmax", [(200, 5000, 6), (3, 60, 1), (70, 100000, 12)])
def test_retrieval_metrics_vs_reference_loop(nq, ndb, gmax):
    from hcir import metrics
    rng = np.random.default_rng(2)
    Ks = (10, 20, 50)
    ret = np.stack([rng.permutation(ndb)[:50] for _ in range(nq)]).astype(np.int64)
    gt = np.full((nq, gmax), -1, dtype=np.int64)
    gt_lists = []
    for i in range(nq):
        n = int(rng.integers(0, gmax + 1))
        # half of the ground truth is planted inside the retrieved list, at random ranks
        ids = list(rng.choice(ret[i], size=n // 2, replace=False)) + list(rng.integers(0, ndb, n - n // 2))
        gt[i, :n] = ids
        gt_lists.append(ids)
    res, hit, ap = metrics.retrieval_metrics(torch.from_numpy(ret).cuda(), torch.from_numpy(gt).cuda(), Ks)
    ref = ometrics.evaluate_ids(ret, gt_lists, Ks)
    assert res["total_queries"] == ref["total_queries"] == nq
    for j, k in enumerate(Ks):
        np.testing.assert_array_equal(hit[j].cpu().numpy(), np.array(ref["hit"][k]))
        np.testing.assert_array_equal(ap[j].cpu().numpy(), np.array(ref["ap"][k]))       # same fp64 operations
        assert abs(res["mAP"][k] - ref["mAP"][k]) <= 1e-12
        assert res["Recall"][k] == ref["Recall"][k]
