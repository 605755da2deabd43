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
