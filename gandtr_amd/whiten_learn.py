"""Learned whitening ("Lw") on the device: produces the ``{"m", "P"}`` that ``cirwhiten`` applies.

Reference (numpy float64 on the CPU): ``whitenlearn(X, qidxs, pidxs)`` (mdir/external/cirtorch/utils/whiten.py:37-70), called by
the ``learn_lw_whitening`` stage with ``values.astype(float64).T`` and the index lists of matching (query, positive) pairs
(mdir/stages/whiten.py:30-75).  Same conventions here: ``X`` is D x N (one descriptor per column), the result is ``(m, P)`` with
``m`` D x 1 and ``P`` D x D float64.  The rows of ``P`` are eigenvectors times ``inv(cholesky(S))`` and therefore defined up to sign
(whitened descriptors differ by a per-dimension sign, scores between them do not).  No CPU fallback."""
import ctypes

import torch

from . import _hip


def whitenlearn(X, qidxs, pidxs, return_info=False):
    """X: D x N tensor on a HIP device (any float dtype; the kernels read float32 rows and compute in float64);
    qidxs / pidxs: equally long sequences (or tensors) of column numbers.  Returns (m [D x 1], P [D x D]) float64 device tensors."""
    lib = _hip.load()
    if not isinstance(X, torch.Tensor) or not X.is_cuda or X.dim() != 2:
        raise ValueError("whitenlearn needs a D x N tensor on a HIP device")
    dev = X.device
    rows = X.t().contiguous().float()                                   # [N][D]
    n_vec, d = rows.shape
    q = torch.as_tensor(qidxs, dtype=torch.int64).reshape(-1)
    p = torch.as_tensor(pidxs, dtype=torch.int64).reshape(-1)
    if q.numel() != p.numel() or q.numel() == 0:
        raise ValueError("query and positive index lists must be non-empty and equally long")
    if int(q.min()) < 0 or int(p.min()) < 0 or int(q.max()) >= n_vec or int(p.max()) >= n_vec:
        raise IndexError("pair index out of range for %d vectors" % n_vec)
    q32, p32 = q.to(dev, torch.int32), p.to(dev, torch.int32)
    m = torch.empty((d,), dtype=torch.float64, device=dev)
    P = torch.empty((d, d), dtype=torch.float64, device=dev)
    eig = torch.empty((d,), dtype=torch.float64, device=dev)
    info = (ctypes.c_int * 2)()
    need = ctypes.c_size_t()
    with torch.cuda.device(dev):
        _hip.check(lib.gdt_whiten_learn_workspace_bytes(n_vec, d, q32.numel(), ctypes.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        _hip.check(lib.gdt_whiten_learn(rows.data_ptr(), q32.data_ptr(), p32.data_ptr(), n_vec, d, q32.numel(), m.data_ptr(),
                                        P.data_ptr(), eig.data_ptr(), info, ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream(dev).cuda_stream))
    if return_info:
        return m[:, None], P, {"eigenvalues": eig, "cholesky_jitter_steps": info[0], "jacobi_sweeps": abs(info[1]),
                               "one_sided": info[1] > 0}        # False: the scatter matrix was singular, two-sided fallback
    return m[:, None], P


def learn_lw_whitening(names, values, queries, positives):
    """Host-side mirror of the ``learn_lw_whitening`` stage body (mdir/stages/whiten.py:30-75): names -> row numbers, then
    ``whitenlearn``.  ``values``: N x D tensor on the device.  Returns {"m": D x 1, "P": D x D} as float64 numpy arrays, the format
    ``CirtorchWhiten`` loads (wrapper.py:315-317)."""
    assert len(names) == len(values)
    assert len(queries) == len(positives)
    index = {x: i for i, x in enumerate(names)}
    q = [index[x] for x in queries]
    p = [index[x] for x in positives]
    m, P = whitenlearn(values.t(), q, p)
    return {"m": m.cpu().numpy(), "P": P.cpu().numpy()}
