"""Generates tests/golden/whiten_learn.npz by running the reference's own whitenlearn (mdir/external/cirtorch/utils/whiten.py, loaded
by path: it needs only os and numpy) on small synthetic descriptor sets.  Run in the build container:
python tests/golden/make_whiten_golden.py"""
import importlib.util
import os

import numpy as np

REF = "/root/reference/mdir/external/cirtorch/utils/whiten.py"


def synth_set(seed, d, n, npairs):
    rng = np.random.default_rng(seed)
    basis = rng.normal(size=(d, d)) * (np.linspace(1.5, 0.2, d)[None, :])          # anisotropic descriptors
    base = rng.normal(size=(n // 2, d)) @ basis.T
    X = np.concatenate([base, base + 0.3 * rng.normal(size=base.shape) @ basis.T])   # second half: noisy positives of the first
    X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
    q = rng.integers(0, n // 2, npairs)
    return X.T.astype(np.float64), q, q + n // 2


def main():
    spec = importlib.util.spec_from_file_location("ref_whiten", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for i, (d, n, npairs) in enumerate([(16, 200, 150), (32, 400, 300)]):
        X, q, p = synth_set(i, d, n, npairs)
        m, P = ref.whitenlearn(X, q, p)
        out.update({"X_%d" % i: X.astype(np.float32), "q_%d" % i: q, "p_%d" % i: p, "m_%d" % i: m, "P_%d" % i: P})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "whiten_learn.npz"), **out)
    print("wrote", len(out) // 5, "cases")


if __name__ == "__main__":
    main()
