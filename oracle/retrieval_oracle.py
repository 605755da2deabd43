"""CPU restatement of the reference's retrieval scoring and cluster-aware hard-negative selection.  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing under gandtr_amd/).

* scores / ranks:      mdir/components/optim/score/cirscore.py:71-73   (numpy dot + argsort)
* negative selection:  mdir/external/cirtorch/datasets/traindataset.py:246-279 (`TuplesDataset._search_hard_negatives`)

Pinning: `tests/golden/hard_negatives.npz` holds outputs of the reference's OWN method -- `TuplesDataset._search_hard_negatives` called unbound on
a namespace carrying `clusters` and `nnum`, the only attributes it reads (tests/golden/make_golden.py, section 9; the class's `__init__` wants the
pickled training database, which the method does not need) -- for four seeded cases (random unit vectors; queries whose own cluster fills the top of
the ranking; few clusters).  `tests/test_oracle_retrieval.py` checks this restatement against them (indices exactly, distances to float32 rounding)
and on hand-made cases whose answer follows from the text of the reference loop.
"""
import numpy as np


def scores_and_ranks(vecs, qvecs):
    """vecs D x Ndb, qvecs D x Nq (numpy) -> scores Ndb x Nq, ranks Ndb x Nq (descending score, stable)"""
    scores = np.dot(vecs.T, qvecs)
    ranks = np.argsort(-scores, axis=0, kind="stable")
    return scores, ranks


def search_hard_negatives(qidxs, qvecs, idxs2images, poolvecs, clusters, nnum, ranks=None):
    """traindataset.py:256-275, statement by statement; returns (nidxs, distances in selection order)"""
    if ranks is None:
        _, ranks = scores_and_ranks(poolvecs, qvecs)
    nidxs, ndist_acc = [], []
    for q in range(len(qidxs)):
        qcluster = clusters[qidxs[q]]                      # :262 do not use query cluster
        used = [qcluster]
        nidx = []
        r = 0
        while len(nidx) < nnum:                            # :266
            potential = idxs2images[ranks[r, q]]           # :267 (IndexError past the pool, as in the reference)
            if clusters[potential] not in used:            # :269 take at most one image from the same cluster
                nidx.append(int(potential))
                used.append(clusters[potential])
                diff = qvecs[:, q] - poolvecs[:, ranks[r, q]] + 1e-6                       # :272
                ndist_acc.append(float(np.sqrt((diff.astype(np.float64) ** 2).sum())))
            r += 1
        nidxs.append(nidx)
    return nidxs, ndist_acc
