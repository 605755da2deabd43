"""oracle/whiten_oracle.py against the golden vectors produced by the reference's own whitenlearn (tests/golden/make_whiten_golden.py),
and the whitening identities that hold at any size."""
import os

import numpy as np

from oracle import whiten_oracle as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whiten_learn.npz")


def test_golden_vectors_from_the_reference():
    g = np.load(GOLD)
    n = sum(1 for k in g.files if k.startswith("X_"))
    assert n >= 2
    for i in range(n):
        m, P, _ = W.whitenlearn(g["X_%d" % i].astype(np.float64), g["q_%d" % i], g["p_%d" % i])
        assert np.array_equal(m, g["m_%d" % i])
        assert W.rows_up_to_sign(P, g["P_%d" % i]) < 1e-9


def test_whitening_identities():
    """P S P^T = I for the pair-difference covariance S, and P C P^T = diag(eigenvalues, decreasing) for the scatter C"""
    rng = np.random.default_rng(0)
    d, n = 24, 300
    X = rng.normal(size=(d, n)) * np.linspace(2, 0.3, d)[:, None]
    q = rng.integers(0, n, 200)
    p = (q + 1 + rng.integers(0, n - 1, 200)) % n
    m, P, w = W.whitenlearn(X, q, p)
    diff = X[:, q] - X[:, p]
    S = diff @ diff.T / diff.shape[1]
    assert np.abs(P @ S @ P.T - np.eye(d)).max() < 1e-9
    C = (X - m) @ (X - m).T
    D = P @ C @ P.T
    assert np.abs(D - np.diag(w)).max() < 1e-7 * w.max()
    assert (np.diff(w) <= 0).all()


def test_jitter_loop_on_a_singular_covariance():
    S = np.zeros((6, 6))
    S[:3, :3] = np.eye(3)
    P0, steps = W.inverse_cholesky(S)
    assert steps >= 1 and np.isfinite(P0).all()
