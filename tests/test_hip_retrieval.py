"""Retrieval scoring on the GPU against the reference's numpy formulation (cirscore.py:71-73)."""
import numpy as np
import pytest
import torch

from gandtr_amd import retrieval
from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu


def _unit(seed, d, n):
    v = synth._normal(seed, "v", (d, n))
    return v / v.norm(dim=0, keepdim=True)


@pytest.mark.parametrize("d,ndb,nq", [(512, 1000, 7), (2048, 4096, 70), (128, 333, 1), (2048, 20000, 55)])
def test_scores_and_ranks_match_numpy(cuda_device, d, ndb, nq):
    vecs, qvecs = _unit(1, d, ndb), _unit(2, d, nq)
    ref_scores = np.dot(vecs.numpy().T.astype(np.float64), qvecs.numpy().astype(np.float64))     # exact reference
    ref32 = np.dot(vecs.numpy().T, qvecs.numpy())                                                # what the reference computes
    scores, ranks = retrieval.scores_and_ranks(vecs.to(cuda_device), qvecs.to(cuda_device))
    scores, ranks = scores.cpu().numpy(), ranks.cpu().numpy()
    assert scores.shape == (ndb, nq) and ranks.shape == (ndb, nq) and ranks.dtype == np.int32
    assert np.abs(scores - ref_scores).max() < 2e-6                       # f16x3 GEMM: fp32-class accuracy
    assert np.abs(scores - ref_scores).max() <= 4 * max(np.abs(ref32 - ref_scores).max(), 2.5e-7)
    for q in range(nq):
        col = ranks[:, q]
        assert sorted(col.tolist()) == list(range(ndb))                   # a permutation of the database
        s = scores[col, q]
        assert np.all(s[:-1] >= s[1:])                                    # ordered by decreasing score
    # same order as numpy wherever the scores are separated by more than the rounding noise
    ref_ranks = np.argsort(-ref_scores, axis=0)
    gaps = np.abs(np.diff(np.take_along_axis(ref_scores, ref_ranks, axis=0), axis=0))
    clear = np.concatenate([gaps > 1e-5, np.ones((1, nq), bool)]) & np.concatenate([np.ones((1, nq), bool), gaps > 1e-5])
    assert np.array_equal(ranks[clear], ref_ranks[clear])


def test_scores_only_and_index_base(cuda_device):
    vecs, qvecs = _unit(3, 256, 500).to(cuda_device), _unit(4, 256, 3).to(cuda_device)
    s, r = retrieval.scores_and_ranks(vecs, qvecs, with_ranks=False)
    assert r is None and s.shape == (500, 3)
    s2, r2 = retrieval.scores_and_ranks(vecs, qvecs, index_base=1000)
    assert torch.equal(s, s2) and int(r2.min()) == 1000 and int(r2.max()) == 1499


def test_invalid_arguments(cuda_device):
    with pytest.raises(ValueError):
        retrieval.scores_and_ranks(torch.zeros(100, 10, device=cuda_device), torch.zeros(100, 2, device=cuda_device))   # D not a power of two
    with pytest.raises(ValueError):
        retrieval.scores_and_ranks(torch.zeros(128, 10, device=cuda_device), torch.zeros(64, 2, device=cuda_device))


@pytest.mark.parametrize("d,npool,nq,nclusters,nnum", [(512, 3000, 40, 200, 5), (2048, 20000, 70, 1500, 5), (128, 400, 9, 12, 10)])
def test_hard_negative_selection_matches_reference_loop(cuda_device, d, npool, nq, nclusters, nnum):
    """``TuplesDataset._search_hard_negatives`` (traindataset.py:246-279) on the device -- mm + sort + the cluster-aware selection -- against the
    CPU restatement of the reference loop run on the DEVICE's ranking (ties in fp32 scores may order differently on two machines; the selection
    itself is integer bookkeeping and must agree exactly), distances to 1e-5"""
    from oracle import retrieval_oracle as R
    rng = np.random.RandomState(d + nq)
    nimg = npool + 500
    poolvecs, qvecs = _unit(3, d, npool), _unit(4, d, nq)
    clusters = rng.randint(0, nclusters, nimg).tolist()
    idxs2images = rng.permutation(nimg)[:npool].tolist()
    qidxs = rng.randint(0, nimg, nq).tolist()
    nidxs, stats = retrieval.search_hard_negatives(qidxs, qvecs.to(cuda_device), idxs2images, poolvecs.to(cuda_device), clusters, nnum)
    _, ranks = retrieval.scores_and_ranks(poolvecs.to(cuda_device), qvecs.to(cuda_device))
    want, wdist = R.search_hard_negatives(qidxs, qvecs.numpy(), idxs2images, poolvecs.numpy(), clusters, nnum, ranks=ranks.cpu().numpy())
    assert nidxs == want
    assert np.allclose(stats["average_negative_distance"], wdist, atol=1e-5)
    for q, row in enumerate(nidxs):                                      # the rules themselves
        cl = [clusters[i] for i in row]
        assert len(set(cl)) == nnum and clusters[qidxs[q]] not in cl


def test_hard_negative_selection_runs_out_of_clusters(cuda_device):
    poolvecs, qvecs = _unit(5, 64, 50), _unit(6, 64, 3)
    with pytest.raises(IndexError):
        retrieval.search_hard_negatives([0, 1, 2], qvecs.to(cuda_device), list(range(10, 60)), poolvecs.to(cuda_device), [0, 1, 2] * 20, 5)


def test_hard_negative_selection_against_the_reference_fixture(cuda_device):
    """`retrieval.search_hard_negatives` (scores, sort and selection on the device) against tests/golden/hard_negatives.npz = outputs of the reference's own
    `TuplesDataset._search_hard_negatives` (traindataset.py:246-279; make_golden.py section 9): the same images in the same order, the same distances"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hard_negatives.npz"))
    for k in range(int(g["cases"])):
        c = {key: g["c%d_%s" % (k, key)] for key in ("poolvecs", "qvecs", "clusters", "idxs2images", "qidxs", "nidxs", "ndist")}
        nidxs, stats = retrieval.search_hard_negatives(c["qidxs"].tolist(), torch.from_numpy(c["qvecs"]).to(cuda_device), c["idxs2images"].tolist(),
                                                       torch.from_numpy(c["poolvecs"]).to(cuda_device), c["clusters"].tolist(), int(g["c%d_nnum" % k]))
        assert nidxs == c["nidxs"].tolist(), k
        assert np.allclose(stats["average_negative_distance"], c["ndist"], rtol=1e-5, atol=1e-6), k
