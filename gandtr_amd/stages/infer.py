"""``infer`` stage -- signature and loop shape of mdir/stages/infer.py:17-66.

The reference body wires datasets (DataLoader with 6 workers), output sinks and resource statistics around
``out = network(indata)``; those subsystems are out of scope, so this mirror accepts in-memory
inputs: ``data[0]`` is a sequence of image tensors (C x H x W or 1 x C x H x W).  Per item it calls ``network(indata)`` under
``torch.no_grad()`` exactly like the reference (batch size 1, the only mode the reference's wrappers support) and collects the
outputs according to ``params["output"]["inference"]["name"]``.  On a HIP device items of EQUAL size are grouped and go through the network as one batch
(at most ``GANDTR_INFER_BATCH`` = 64 items AND at most ``GANDTR_INFER_PIXELS`` pixels per batch -- default 64 x 256 x 256 for image outputs, 32 x 1024 x 1024 for
embeddings: the benchmarked geometries, so 1024 x 1024 generator inputs go four at a time and the workspace stays what it is for those; a batch the device
cannot allocate is halved and retried; 0 / 1 = the reference's item-by-item loop): no op of either model family crosses images (InstanceNorm is per image,
BatchNorm is in eval mode, GeM / L2N / whitening per image), so the outputs are those of the loop in the input's order -- WITHIN THE PATH'S TOLERANCE, not bit
for bit: a batch selects other kernel forms (patch kernels instead of the small-launch ones) than a single image does (generator pre-tanh 1e-3 relative,
descriptors |d| <= 1e-3 / cosine >= 0.9999 against the fp32 oracle either way) -- while the conv kernels run at 6-9x the batch-1 rate.  An
embedding item that is a batch already (N > 1) sends the whole call through the item-by-item loop, as the reference would run it.  Outputs:
    "embedding" -> one (N x D) float32 numpy array   (EmbeddingOutput, mdir/components/data/output.py:118-156)
    "rgb"       -> a list of H x W x 3 float arrays in [0, 1] (un-normalised with the network's mean_std; RgbImageSaver :75-84)
"""
import copy
import os
import time

import numpy as np
import torch

from ..learning import load_network


def infer(params, data):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    np.random.seed(0)
    torch.manual_seed(0)
    if not len(data[0]):
        return ({"status": "skipped"},)
    network = load_network(copy.deepcopy(params["network"]), device).eval()
    kind = params["output"]["inference"]["name"]
    if kind not in ("embedding", "rgb"):
        raise KeyError(kind)
    mean_std = network.network_params.runtime.get("data", {}).get("mean_std")
    outputs, t0 = [], time.time()
    max_batch = int(os.environ.get("GANDTR_INFER_BATCH", "64")) if device.type == "cuda" else 1
    if max_batch > 1 and "forward" not in params:
        singles = all(torch.as_tensor(x).dim() == 3 or torch.as_tensor(x).shape[0] == 1 for x in data[0])
        if kind == "rgb" or singles:                        # (an embedding item with N > 1: the loop below, extract_vectors takes one image per entry)
            return _infer_grouped(network, data[0], kind, mean_std, device, max_batch, t0)
    with torch.no_grad():
        forward = getattr(network, params["forward"]["method"]) if "forward" in params else network
        for indata in data[0]:
            indata = torch.as_tensor(indata)
            if indata.dim() == 3:
                indata = indata.unsqueeze(0)
            if "forward" in params:
                out = forward(indata.to(device), **params["forward"]["params"])
            else:
                out = network(indata)
            if kind == "embedding":
                outputs.append(out.detach().float().cpu().numpy().reshape(-1))
            else:
                img = out.detach().float().cpu()[0]
                if mean_std is not None:
                    img = img * torch.tensor(mean_std[1])[:, None, None] + torch.tensor(mean_std[0])[:, None, None]
                outputs.append(img.clamp(0, 1).permute(1, 2, 0).numpy())
    metadata = {"stats": {"items": len(outputs), "seconds": time.time() - t0}}
    if kind == "embedding":
        return (metadata, np.stack(outputs))
    return (metadata, outputs)


def _pixel_budget(kind):
    default = 32 * 1024 * 1024 if kind == "embedding" else 64 * 256 * 256
    return max(1, int(os.environ.get("GANDTR_INFER_PIXELS", default)))


def _is_allocation_failure(err):
    text = str(err).lower()
    return isinstance(err, torch.cuda.OutOfMemoryError) or "out of memory" in text or "hipmalloc" in text


def _infer_grouped(network, items, kind, mean_std, device, max_batch, t0):
    """equal-size items as batches, outputs in input order (see the module docstring)"""
    from .validate import extract_vectors
    xs = []
    for indata in items:
        x = torch.as_tensor(indata)
        xs.append(x.unsqueeze(0) if x.dim() == 3 else x)
    budget = _pixel_budget(kind)
    if kind == "embedding":
        vecs = extract_vectors(network, xs, device, batched=True, max_batch=min(max_batch, 32), max_pixels=budget)        # D x N on the device
        return ({"stats": {"items": len(xs), "seconds": time.time() - t0}}, vecs.t().contiguous().cpu().numpy())
    outputs = [None] * len(xs)
    jobs, groups = [], {}
    for i, x in enumerate(xs):
        if x.shape[0] != 1:
            jobs.append([i])                                # an item that is a batch already: the loop takes its first image, so does this path
        else:
            groups.setdefault(tuple(x.shape), []).append(i)
    for shape, idx in groups.items():
        step = max(1, min(max_batch, budget // max(1, shape[-1] * shape[-2])))
        jobs.extend(idx[lo:lo + step] for lo in range(0, len(idx), step))
    mean = torch.tensor(mean_std[0], device=device)[:, None, None] if mean_std is not None else None
    std = torch.tensor(mean_std[1], device=device)[:, None, None] if mean_std is not None else None
    with torch.no_grad():
        jobs.reverse()                                      # (a work stack: a batch the device cannot allocate comes back as its two halves)
        while jobs:
            part = jobs.pop()
            try:
                out = network(torch.cat([xs[i] for i in part], 0) if len(part) > 1 else xs[part[0]]).detach().float()
            except (RuntimeError, MemoryError) as err:
                if len(part) == 1 or not _is_allocation_failure(err):
                    raise
                torch.cuda.empty_cache()
                jobs.extend([part[len(part) // 2:], part[:len(part) // 2]])
                continue
            if mean is not None:
                out = out * std + mean
            img = out.clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
            for j, i in enumerate(part):
                outputs[i] = img[j]
    return ({"stats": {"items": len(outputs), "seconds": time.time() - t0}}, outputs)
