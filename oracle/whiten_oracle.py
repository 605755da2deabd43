"""CPU oracle for learned whitening -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates ``whitenlearn`` / ``cholesky`` of mdir/external/cirtorch/utils/whiten.py:37-70 (numpy float64): mean of the query vectors,
covariance of the (query - positive) differences, inverse Cholesky factor with a growing diagonal jitter until the factorisation
succeeds, eigen-decomposition of the projected scatter of all vectors, eigenvalues in decreasing order, ``P = eigvec^T P0``.
Pinned: tests/test_oracle_whiten.py loads the reference's own whiten.py by path (it imports only os and numpy) and compares results
bit for bit in the build container; tests/golden/whiten_learn.npz holds vectors generated the same way
(tests/golden/make_whiten_golden.py).  Eigenvectors carry an arbitrary sign: comparisons of P are per row up to sign.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np


def inverse_cholesky(S):
    """P0 = inv(L) with L L^T = S + alpha I, alpha = 0, 1e-10, 1e-9, ... (whiten.py:55-70); returns (P0, number of jitter steps)"""
    alpha, steps = 0.0, 0
    while True:
        try:
            L = np.linalg.cholesky(S + alpha * np.eye(S.shape[0]))
            return np.linalg.inv(L), steps
        except np.linalg.LinAlgError:
            alpha = 1e-10 if alpha == 0 else alpha * 10
            steps += 1


def whitenlearn(X, qidxs, pidxs):
    """X: D x N float64; returns (m [D x 1], P [D x D], eigenvalues in decreasing order)"""
    X = np.asarray(X, np.float64)
    m = X[:, qidxs].mean(axis=1, keepdims=True)
    diff = X[:, qidxs] - X[:, pidxs]
    P0, _ = inverse_cholesky(diff @ diff.T / diff.shape[1])
    Y = P0 @ (X - m)
    w, U = np.linalg.eig(Y @ Y.T)
    order = np.argsort(w)[::-1]
    return m, U[:, order].T @ P0, w[order]


def rows_up_to_sign(P, Q):
    """max over rows of min(|P_i - Q_i|, |P_i + Q_i|) relative to |Q_i| (eigenvector sign ambiguity)"""
    a = np.linalg.norm(P - Q, axis=1)
    b = np.linalg.norm(P + Q, axis=1)
    return float((np.minimum(a, b) / np.linalg.norm(Q, axis=1)).max())
