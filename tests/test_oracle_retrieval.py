"""The CPU restatement of the reference's hard-negative selection (oracle/retrieval_oracle.py) against the fixture the reference's own
`TuplesDataset._search_hard_negatives` produced (tests/golden/hard_negatives.npz, make_golden.py section 9), on cases whose answer follows from the
text of traindataset.py:256-275, and against a torch transcription of that loop (torch.mm / torch.sort, as the reference writes it)."""
import os

import numpy as np
import pytest
import torch

from oracle import retrieval_oracle as R


def _reference_loop_torch(qidxs, qvecs, idxs2images, poolvecs, clusters, nnum):
    """the reference's statements with torch ops (traindataset.py:250-275), minus the printing"""
    scores = torch.mm(poolvecs.t(), qvecs)
    scores, ranks = torch.sort(scores, dim=0, descending=True, stable=True)
    nidxs = []
    for q in range(len(qidxs)):
        qcluster = clusters[qidxs[q]]
        used = [qcluster]
        nidx = []
        r = 0
        while len(nidx) < nnum:
            potential = idxs2images[ranks[r, q]]
            if not clusters[potential] in used:
                nidx.append(potential)
                used.append(clusters[potential])
            r += 1
        nidxs.append(nidx)
    return nidxs


def test_hand_case_order_and_cluster_rules():
    # one query (image 0, cluster 7); pool positions 0..5 are images 10..15 with clusters 7 3 3 4 7 5; scores descending by position
    qvecs = np.array([[1.0], [0.0]], dtype=np.float32)
    pool = np.array([[0.9, 0.8, 0.7, 0.6, 0.5, 0.4], [0, 0, 0, 0, 0, 0]], dtype=np.float32)
    clusters = {0: 7, 10: 7, 11: 3, 12: 3, 13: 4, 14: 7, 15: 5}
    nidxs, dist = R.search_hard_negatives([0], qvecs, [10, 11, 12, 13, 14, 15], pool, clusters, 3)
    assert nidxs == [[11, 13, 15]]            # 10 / 14 share the query's cluster, 12 repeats cluster 3
    assert np.allclose(dist, [np.sqrt((0.2 + 1e-6) ** 2 + 1e-12), np.sqrt((0.4 + 1e-6) ** 2 + 1e-12), np.sqrt((0.6 + 1e-6) ** 2 + 1e-12)], rtol=1e-5)
    with pytest.raises(IndexError):
        R.search_hard_negatives([0], qvecs, [10, 11, 12, 13, 14, 15], pool, clusters, 4)      # only three other clusters exist


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_against_torch_transcription_of_the_loop(seed):
    rng = np.random.RandomState(seed)
    d, npool, nq, nimg = 32, 300, 17, 400
    poolvecs = rng.randn(d, npool).astype(np.float32); poolvecs /= np.linalg.norm(poolvecs, axis=0)
    qvecs = rng.randn(d, nq).astype(np.float32); qvecs /= np.linalg.norm(qvecs, axis=0)
    clusters = rng.randint(0, 40, nimg).tolist()
    idxs2images = rng.permutation(nimg)[:npool].tolist()
    qidxs = rng.randint(0, nimg, nq).tolist()
    want = _reference_loop_torch(qidxs, torch.from_numpy(qvecs), idxs2images, torch.from_numpy(poolvecs), clusters, 5)
    got, dist = R.search_hard_negatives(qidxs, qvecs, idxs2images, poolvecs, clusters, 5)
    assert got == want and len(dist) == nq * 5


def _fixture():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hard_negatives.npz"))


def test_against_the_reference_methods_own_output():
    """the fixture: outputs of the reference's `_search_hard_negatives` itself (called unbound, make_golden.py section 9)"""
    g = _fixture()
    n = int(g["cases"])
    assert n >= 3
    skipped_own_cluster = 0
    for k in range(n):
        c = {key: g["c%d_%s" % (k, key)] for key in ("poolvecs", "qvecs", "clusters", "idxs2images", "qidxs", "nidxs", "ndist")}
        nnum = int(g["c%d_nnum" % k])
        got, dist = R.search_hard_negatives(c["qidxs"].tolist(), c["qvecs"], c["idxs2images"].tolist(), c["poolvecs"], c["clusters"].tolist(), nnum)
        assert got == c["nidxs"].tolist(), k
        assert np.allclose(dist, c["ndist"], rtol=2e-6, atol=1e-7), k           # (the reference sums the squares in float32)
        # the case is worth its name: count the top-ranked candidates the rules had to skip
        _, ranks = R.scores_and_ranks(c["poolvecs"], c["qvecs"])
        for q in range(len(c["qidxs"])):
            skipped_own_cluster += int(c["clusters"][c["idxs2images"][ranks[0, q]]] == c["clusters"][c["qidxs"][q]])
    assert skipped_own_cluster >= 6          # case 1's first six queries: the best-scoring pool image belongs to the query's own cluster
