"""Learned whitening on the GPU (gandtr_amd/csrc/whiten_learn.hip through the C ABI) against the reference's golden vectors and the
numpy oracle.  float64 throughout; P is compared per row up to sign (eigenvectors), plus the sign-free identities P S P^T = I and
P C P^T = diag(decreasing eigenvalues), which also hold at the full descriptor size."""
import os

import numpy as np
import pytest
import torch

from gandtr_amd import whiten_learn
from oracle import whiten_oracle as W

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whiten_learn.npz")


def test_golden_vectors_from_the_reference(cuda_device):
    g = np.load(GOLD)
    n = sum(1 for k in g.files if k.startswith("X_"))
    for i in range(n):
        X = torch.from_numpy(g["X_%d" % i]).to(cuda_device)
        m, P = whiten_learn.whitenlearn(X, g["q_%d" % i], g["p_%d" % i])
        assert m.shape == g["m_%d" % i].shape and P.dtype == torch.float64
        assert np.abs(m.cpu().numpy() - g["m_%d" % i]).max() < 1e-14
        assert W.rows_up_to_sign(P.cpu().numpy(), g["P_%d" % i]) < 1e-8, i


def _descriptors(seed, d, n):
    rng = np.random.default_rng(seed)
    basis = rng.normal(size=(d, d)) * np.linspace(1.5, 0.2, d)[None, :]
    base = rng.normal(size=(n // 2, d)) @ basis.T
    X = np.concatenate([base, base + 0.3 * rng.normal(size=base.shape) @ basis.T])
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    q = rng.integers(0, n // 2, n)
    return X.T, q, q + n // 2


@pytest.mark.parametrize("d,n", [(64, 1000), (128, 3000), (130, 700)])
def test_matches_oracle_up_to_sign(cuda_device, d, n):
    X, q, p = _descriptors(d, d, n)
    m_ref, P_ref, w_ref = W.whitenlearn(X.astype(np.float64), q, p)
    m, P, info = whiten_learn.whitenlearn(torch.from_numpy(X).to(cuda_device), q, p, return_info=True)
    assert np.abs(m.cpu().numpy() - m_ref).max() < 1e-13
    w = info["eigenvalues"].cpu().numpy()
    assert np.abs(w - w_ref).max() < 1e-9 * w_ref.max()
    assert W.rows_up_to_sign(P.cpu().numpy(), P_ref) < 1e-7
    assert info["cholesky_jitter_steps"] == 0 and 1 <= info["jacobi_sweeps"] <= 30 and info["one_sided"]
    # what the consumer sees: whitened descriptors agree up to a per-dimension sign, scores between them exactly
    Xd = X.astype(np.float64)
    a = P.cpu().numpy() @ (Xd - m.cpu().numpy())
    b = P_ref @ (Xd - m_ref)
    assert np.abs(a.T @ a - b.T @ b).max() < 1e-6 * np.abs(b.T @ b).max()


def test_identities_at_descriptor_size_512(cuda_device):
    d, n = 512, 6000
    X, q, p = _descriptors(7, d, n)
    m, P, info = whiten_learn.whitenlearn(torch.from_numpy(X).to(cuda_device), q, p, return_info=True)
    P, m, w = P.cpu().numpy(), m.cpu().numpy(), info["eigenvalues"].cpu().numpy()
    Xd = X.astype(np.float64)
    diff = Xd[:, q] - Xd[:, p]
    S = diff @ diff.T / diff.shape[1]
    assert np.abs(P @ S @ P.T - np.eye(d)).max() < 1e-8                    # whitens the pair differences
    C = (Xd - m) @ (Xd - m).T
    D = P @ C @ P.T
    assert np.abs(D - np.diag(w)).max() < 1e-8 * w.max()                    # and diagonalises the projected scatter
    assert (np.diff(w) <= 0).all()
    again = whiten_learn.whitenlearn(torch.from_numpy(X).to(cuda_device), q, p)[1]
    assert torch.equal(again.cpu(), torch.from_numpy(P))                    # deterministic


def test_identities_at_full_descriptor_size_2048(cuda_device):
    """GeM-ResNet-101 descriptor size: the sign-free identities, checked on the device in float64"""
    d, n = 2048, 12000
    g = torch.Generator(device=cuda_device).manual_seed(0)
    X = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=cuda_device) *
                                      torch.linspace(1.5, 0.2, d, device=cuda_device)[None, :], dim=1)
    q = torch.randint(0, n // 2, (6000,), generator=torch.Generator().manual_seed(1))
    p = q + n // 2
    m, P, info = whiten_learn.whitenlearn(X.t(), q, p, return_info=True)
    assert info["one_sided"] and info["cholesky_jitter_steps"] == 0
    Xd = X.double()
    diff = Xd[q.to(cuda_device)] - Xd[p.to(cuda_device)]
    S = diff.t() @ diff / len(q)
    eye = torch.eye(d, device=cuda_device, dtype=torch.float64)
    assert float((P @ S @ P.t() - eye).abs().max()) < 1e-8
    Xc = Xd - m.t()
    D = P @ (Xc.t() @ Xc) @ P.t()
    w = info["eigenvalues"]
    assert float((D - torch.diag(w)).abs().max()) < 1e-8 * float(w.max())
    assert bool((w[1:] <= w[:-1]).all())


def test_jitter_and_argument_errors(cuda_device):
    X = torch.zeros(8, 40)
    X[:4] = torch.randn(4, 40, generator=torch.Generator().manual_seed(0))  # rank-deficient covariance: needs the diagonal jitter
    q, p = list(range(0, 20)), list(range(20, 40))
    m, P, info = whiten_learn.whitenlearn(X.to(cuda_device), q, p, return_info=True)
    assert info["cholesky_jitter_steps"] >= 1 and torch.isfinite(P).all() and not info["one_sided"]      # singular scatter: fallback path
    Xd = X.double().numpy()
    m_ref, P_ref, w_ref = W.whitenlearn(Xd, q, p)
    w = info["eigenvalues"].cpu().numpy()
    assert np.abs(w[:4] - w_ref[:4].real).max() < 1e-6 * abs(w_ref[0])                                    # the four non-zero directions
    with pytest.raises(ValueError):
        whiten_learn.whitenlearn(X.to(cuda_device), [0, 1], [2])
    with pytest.raises(IndexError):
        whiten_learn.whitenlearn(X.to(cuda_device), [0, 99], [1, 2])
    with pytest.raises(ValueError):
        whiten_learn.whitenlearn(torch.zeros(7, 40).to(cuda_device), q, p)                      # odd descriptor size
    out = whiten_learn.learn_lw_whitening(["a%d" % i for i in range(40)], X.t().contiguous().to(cuda_device),
                                          ["a0", "a1"], ["a20", "a21"])
    assert out["P"].shape == (8, 8) and out["m"].shape == (8, 1) and out["P"].dtype == np.float64


def test_stages_learn_then_apply(cuda_device):
    """learn_lw_whitening -> whiten (mdir/stages/whiten.py): host formats in and out, numpy reference for the apply step"""
    import mdir.stages.whiten as stage
    X, q, p = _descriptors(3, 64, 800)
    names = ["im%d" % i for i in range(X.shape[1])]
    values = X.T.copy()                                                         # N x D, as the stage receives them
    meta, lw = stage.learn_lw_whitening({}, (names, values, [names[i] for i in q], [names[i] for i in p]))
    assert meta["stats"] == {"failed_times": 0, "vectors_used": 1.0, "vectors_total": len(q)} and "whitening_learn" in meta["timings"]
    assert lw["m"].shape == (64, 1) and lw["P"].shape == (64, 64) and lw["P"].dtype == np.float64
    m_ref, P_ref, _ = W.whitenlearn(X.astype(np.float64), q, p)
    assert W.rows_up_to_sign(lw["P"], P_ref) < 1e-7
    meta, names2, out = stage.whiten({"dimensions": 32}, (lw, names, values))
    assert names2 == names and out.shape == (800, 32)
    ref = lw["P"][:32] @ (X.astype(np.float64) - lw["m"])                       # whitenapply, cirtorch/utils/whiten.py:4-12
    ref = (ref / (np.linalg.norm(ref, axis=0, keepdims=True) + 1e-6)).T
    assert out.dtype == np.float64 and np.abs(out - ref).max() < 1e-12        # float64 on the device like numpy on the host
