"""``validate`` stage -- the retrieval-evaluation caller of the hot path, with the signature and loop shape of
mdir/stages/validate.py:15-39 -> CirDatasetAp.__call__ (mdir/components/optim/score/cirscore.py:51-73) ->
cirtorch ``extract_vectors`` (mdir/external/cirtorch/networks/imageretrievalnet.py:312-339).

The reference wires dataset files, a DataLoader (6 workers, batch size 1) and the mAP evaluation around three lines of arithmetic:
    vecs[:, i] = net(input).cpu().data.squeeze()         for every database / query image (batch 1)
    scores = np.dot(vecs.T, qvecs)
    ranks  = np.argsort(-scores, axis=0)
Datasets, ground-truth files and ``compute_map`` are out of scope; this mirror takes the images
in memory -- ``data[0]`` database images, ``data[1]`` query images (omitted: queries = database, the ``self.images == self.qimages``
branch, cirscore.py:58-59) -- and returns what the evaluation consumes: ``(metadata, ranks, scores)``, ranks Ndb x Nq database indices
per query column, best first.  On a HIP device the descriptors never leave the GPU between extraction and ranking
(gandtr_amd/retrieval.py: split-fp16 GEMM + segmented radix sort); on the CPU the reference's two numpy lines run as they are.
"""
import copy
import time

import numpy as np
import torch

from ..learning import load_network


def extract_vectors(net, images, device=None):
    """D x N descriptor matrix of a list of image tensors (C x H x W or 1 x C x H x W), one forward per image like the reference's
    batch-size-1 loader loop (the hub's multi-scale / whitening wrappers only support batch 1).  Stays on ``device``."""
    device = torch.device(device) if device is not None else getattr(net, "device", torch.device("cpu"))
    net.eval()
    cols = []
    with torch.no_grad():
        for img in images:
            x = torch.as_tensor(img)
            if x.dim() == 3:
                x = x.unsqueeze(0)
            cols.append(net(x.to(device)).detach().float().reshape(-1))
    dim = net.meta["out_channels"] if getattr(net, "meta", None) and "out_channels" in net.meta else (cols[0].numel() if cols else 0)
    if not cols:
        return torch.zeros((dim, 0), device=device)
    return torch.stack(cols, dim=1)


def rank(vecs, qvecs):
    """(scores Ndb x Nq float32, ranks Ndb x Nq int) as numpy arrays: cirscore.py:71-73."""
    if vecs.is_cuda:
        from .. import retrieval
        scores, ranks = retrieval.scores_and_ranks(vecs, qvecs)
        return scores.cpu().numpy(), ranks.cpu().numpy().astype(np.int64)
    v, q = vecs.numpy(), qvecs.numpy()
    scores = np.dot(v.T, q)
    return scores, np.argsort(-scores, axis=0)


def validate(params, data):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    np.random.seed(0)
    torch.manual_seed(0)
    assert params.keys() == {"network", "validation", "data"}, params.keys()
    network = load_network(copy.deepcopy(params["network"]), device).eval()
    images = data[0]
    qimages = data[1] if len(data) > 1 and data[1] is not None else None
    t0 = time.time()
    vecs = extract_vectors(network, images, device)
    qvecs = vecs.clone() if qimages is None else extract_vectors(network, qimages, device)
    t1 = time.time()
    scores, ranks = rank(vecs, qvecs)
    t2 = time.time()
    metadata = {"eval": {"database": vecs.shape[1], "queries": qvecs.shape[1], "dim": vecs.shape[0],
                         "extract_descriptors_s": t1 - t0, "compute_score_s": t2 - t1}}
    return (metadata, ranks, scores)
