"""``whiten`` and ``learn_lw_whitening`` stages -- signatures and bookkeeping of mdir/stages/whiten.py:10-75, the descriptor
arithmetic on the device (``gdt_whiten`` / ``gdt_whiten_learn``).  Inputs and outputs keep the reference's host formats: ``values``
is an N x D float array, the learned whitening is ``{"m": D x 1, "P": D x D}`` float64 numpy (what ``cirwhiten`` loads,
wrapper.py:315-317).  ``resource_usage`` of the reference's metadata comes from its statistics subsystem (out of scope) and is
omitted.  There is no CPU fallback: without a HIP device these stages raise."""
import time

import numpy as np
import torch

from .. import engine, whiten_learn


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("the whitening stages run on a HIP device; none is available")
    return torch.device("cuda")


def whiten(params, data):
    """Apply pre-computed whitening (mdir/stages/whiten.py:10-27; whitenapply, cirtorch/utils/whiten.py:4-12).  The reference
    multiplies in float64 on the host (numpy promotes against the float64 P): so does the device path here (``gdt_whiten_f64``);
    the float32 ``gdt_whiten`` is the arithmetic of the ``cirwhiten`` wrapper of the inference path (wrapper.py:320-322)."""
    dimensions = params.pop("dimensions", None) or None
    assert not params, params.keys()
    whitening, names, values = data
    assert len(names) == len(values)
    if not whitening:
        return {"status": "No whitening applied"}, names, values
    dev = _device()
    time0 = time.time()
    v = torch.as_tensor(np.asarray(values), dtype=torch.float64, device=dev)                       # N x D
    P = torch.as_tensor(np.asarray(whitening["P"]), dtype=torch.float64, device=dev)
    m = torch.as_tensor(np.asarray(whitening["m"]), dtype=torch.float64, device=dev).reshape(-1)
    whitened = engine.whiten(v, P, m, dimensions, float64=True).cpu().numpy()                      # N x dims, float64 like the reference
    return {"timings": {"whitening_apply": round(time.time() - time0, 2)}}, names, whitened


def learn_lw_whitening(params, data):
    """Learn Lw whitening (mdir/stages/whiten.py:30-75).  The reference retries on a shuffled subset when numpy reports a matrix
    that is not positive definite; its ``cholesky`` helper already absorbs that case with a growing diagonal jitter
    (cirtorch/utils/whiten.py:55-70), as does the device implementation, so the first trial always succeeds here."""
    assert not params
    names, values, queries, positives = data
    assert len(names) == len(values)
    assert len(queries) == len(positives)
    if not len(names) and not len(queries):
        return {"status": "Empty whitening produced"}, None
    dev = _device()
    name_index = {x: i for i, x in enumerate(names)}
    qidxs = np.array([name_index[x] for x in queries])
    pidxs = np.array([name_index[x] for x in positives])
    time0 = time.time()
    x = torch.as_tensor(np.asarray(values), dtype=torch.float32, device=dev)                       # N x D
    whit_m, whit_p = whiten_learn.whitenlearn(x.t(), qidxs, pidxs)
    timing = time.time() - time0
    metadata = {"stats": {"failed_times": 0, "vectors_used": 1.0, "vectors_total": len(qidxs)},
                "timings": {"whitening_learn": round(timing, 2)}}
    return metadata, {"m": whit_m.cpu().numpy(), "P": whit_p.cpu().numpy()}
