"""``validate`` stage -- the retrieval-evaluation caller of the hot path, with the signature and loop shape of
mdir/stages/validate.py:15-39 -> CirDatasetAp.__call__ (mdir/components/optim/score/cirscore.py:51-73) ->
cirtorch ``extract_vectors`` (mdir/external/cirtorch/networks/imageretrievalnet.py:312-339).

The reference wires dataset files, a DataLoader (6 workers, batch size 1) and the mAP evaluation around three lines of arithmetic:
    vecs[:, i] = net(input).cpu().data.squeeze()         for every database / query image (batch 1)
    scores = np.dot(vecs.T, qvecs)
    ranks  = np.argsort(-scores, axis=0)
Datasets, ground-truth files and ``compute_map`` are out of scope; this mirror takes the images
in memory -- ``data[0]`` database images, ``data[1]`` query images (omitted: queries = database, the ``self.images == self.qimages``
branch, cirscore.py:58-59).  ``validate`` keeps the reference's return arity ``(metadata,)`` (ranks / scores inside the metadata) and
rejects the dataset / criterion parameters it cannot honour; ``rank_images`` returns ``(metadata, ranks, scores)``, ranks Ndb x Nq
database indices per query column, best first.  On a HIP device the descriptors never leave the GPU between extraction and ranking
(gandtr_amd/retrieval.py: split-fp16 GEMM + segmented radix sort); on the CPU the reference's two numpy lines run as they are.
"""
import copy
import os
import time

import numpy as np
import torch

from ..learning import load_network


def extract_vectors(net, images, device=None, batched=None, max_batch=32, concurrent=8, max_pixels=None):
    """D x N descriptor matrix of a list of image tensors (C x H x W or 1 x C x H x W); stays on ``device``.

    The reference runs one forward per image (batch-size-1 DataLoader, imageretrievalnet.py:319-333) because image sizes differ and its
    multi-scale / whitening wrappers only support batch 1.  Here images of EQUAL size are grouped, each group goes
    through the network as one batch of at most ``max_batch`` (the wrappers return the stack of the per-image results: D x n, one
    column per image), and the columns land at the images' positions in the input list -- same result, output order unchanged.
    The conv kernels are 3-4x more efficient at batch >= 8 than at batch 1.  ``batched=None``: on for a HIP device, off on the CPU
    (there the loop is the reference's own arithmetic, image by image); ``batched=False`` forces the reference's loop.  Groups of at most four images (sizes
    that occur rarely) go to the network together, up to ``concurrent`` images in flight (``SingleNetwork.forward_list``: one forward per group and pyramid
    level, all issued before the first is joined; 48 images of 48 sizes through the multi-scale ResNet-101: 264 desc/s image by image, 388 with eight in flight).
    ``max_pixels`` bounds a batch by N x H x W as well (the workspace grows with it); a batch the device cannot allocate is halved and retried."""
    device = torch.device(device) if device is not None else getattr(net, "device", torch.device("cpu"))
    if batched is None:
        batched = device.type == "cuda"
    net.eval()
    items = []
    for img in images:
        x = torch.as_tensor(img)
        items.append(x.unsqueeze(0) if x.dim() == 3 else x)
    dim = net.meta["out_channels"] if getattr(net, "meta", None) and "out_channels" in net.meta else None
    if not items:
        return torch.zeros((dim or 0, 0), device=device)
    cols = [None] * len(items)
    with torch.no_grad():
        if not batched:
            for i, x in enumerate(items):
                cols[i] = net(x.to(device)).detach().float().reshape(-1)
        else:
            groups = {}
            for i, x in enumerate(items):
                assert x.shape[0] == 1, "one image per list entry, got %s" % (tuple(x.shape),)
                groups.setdefault(tuple(x.shape[1:]), []).append(i)
            jobs = []
            for shape, idx in groups.items():                       # dict order = first appearance: deterministic
                step = max_batch if max_pixels is None else max(1, min(max_batch, int(max_pixels) // max(1, shape[-1] * shape[-2])))
                for lo in range(0, len(idx), step):
                    jobs.append(idx[lo:lo + step])
            # small jobs (sizes that occur once or twice) are handed over together, up to ``concurrent`` images: their forwards run side by side on the device
            many = getattr(net, "forward_list", None)
            at = 0
            while at < len(jobs):
                take, inflight = 1, len(jobs[at])
                if many is not None and concurrent > 1 and len(jobs[at]) <= 4:
                    while at + take < len(jobs) and len(jobs[at + take]) <= 4 and inflight + len(jobs[at + take]) <= concurrent:
                        inflight += len(jobs[at + take])
                        take += 1
                parts = jobs[at:at + take]
                batches = [torch.cat([items[i].to(device) for i in part], 0) for part in parts]
                try:
                    outs = many(batches) if take > 1 else [net(batches[0])]
                except (RuntimeError, MemoryError) as err:          # a batch the device cannot allocate: its halves take its place in the job list
                    text = str(err).lower()
                    big = max(range(take), key=lambda k: len(parts[k]))
                    if len(parts[big]) == 1 or not (isinstance(err, torch.cuda.OutOfMemoryError) or "out of memory" in text or "hipmalloc" in text):
                        raise
                    del batches
                    torch.cuda.empty_cache()
                    half = len(parts[big]) // 2
                    jobs[at + big:at + big + 1] = [parts[big][:half], parts[big][half:]]
                    continue
                for part, out in zip(parts, outs):
                    out = out.detach().float().reshape(-1, len(part))   # (D,) for one image, D x n otherwise
                    for j, i in enumerate(part):
                        cols[i] = out[:, j]
                at += take
    return torch.stack(cols, dim=1)


def extract_vectors_from_files(net, files, image_size, mean_std, device=None, host_loader=None, clahe_clip=None, max_batch=32, chunk=64):
    """``extract_vectors(net, images, image_size, transform)`` as the reference calls it -- on image FILES (imageretrievalnet.py:312-339
    builds ``ImagesFromList(root='', images=images, imsize=image_size, transform=transform)`` and a batch-1 loader over it).  ``files``:
    paths or file contents.  Per chunk of ``chunk`` files: JPEG decoding, ``imresize`` and ``totensor | normalize`` (``mean_std``; with
    ``clahe_clip`` the hub's ``apply_clahe`` step in between) run on the device (gandtr_amd/jpeg.py, ingest.py -- bit-identical pixels to
    Pillow's), then ``extract_vectors`` batches the equal-sized ones.  Files the device decoder does not take go to ``host_loader``
    (bytes -> H x W x 3 uint8; the reference's pil_loader) or raise.  Returns D x N on ``device``."""
    from .. import jpeg
    device = torch.device(device) if device is not None else getattr(net, "device", None)
    if device is None or torch.device(device).type != "cuda":
        raise ValueError("extract_vectors_from_files decodes on a HIP device; pass decoded tensors to extract_vectors on the CPU")
    mean, std = mean_std
    cols = []
    for lo in range(0, len(files), chunk):
        tensors = jpeg.ingest_files(files[lo:lo + chunk], image_size, mean, std, clahe_clip=clahe_clip, device=device, host_loader=host_loader)
        cols.append(extract_vectors(net, tensors, device, batched=True, max_batch=max_batch))
    if not cols:
        return extract_vectors(net, [], device)
    return torch.cat(cols, dim=1)


def rank(vecs, qvecs):
    """(scores Ndb x Nq float32, ranks Ndb x Nq int) as numpy arrays: cirscore.py:71-73."""
    if vecs.is_cuda:
        from .. import retrieval
        scores, ranks = retrieval.scores_and_ranks(vecs, qvecs)
        return scores.cpu().numpy(), ranks.cpu().numpy().astype(np.int64)
    v, q = vecs.numpy(), qvecs.numpy()
    scores = np.dot(v.T, q)
    return scores, np.argsort(-scores, axis=0)


def _rank_images(params, data):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    np.random.seed(0)
    torch.manual_seed(0)
    network = load_network(copy.deepcopy(params["network"]), device).eval()
    images = data[0]
    qimages = data[1] if len(data) > 1 and data[1] is not None else None

    def describe(items):
        # image FILES (paths or contents), as the reference's datasets hold them: decoded, resized to the dataset's image size (cirtorch's test
        # size 1024 unless data[2] = {"image_size": s} says otherwise) and normalised with the network's own mean / std on the device
        if len(items) and isinstance(items[0], (str, bytes, bytearray, os.PathLike)):
            opts = data[2] if len(data) > 2 and data[2] else {}
            mean_std = network.network_params.runtime.get("data", {}).get("mean_std")
            if mean_std is None:
                raise ValueError("image files need runtime.data.mean_std in the network parameters (the transform the reference builds its dataset with)")
            return extract_vectors_from_files(network, list(items), opts.get("image_size", 1024), mean_std, device, host_loader=opts.get("host_loader"))
        return extract_vectors(network, items, device)

    t0 = time.time()
    vecs = describe(images)
    qvecs = vecs.clone() if qimages is None else describe(qimages)
    t1 = time.time()
    scores, ranks = rank(vecs, qvecs)
    t2 = time.time()
    metadata = {"eval": {"database": vecs.shape[1], "queries": qvecs.shape[1], "dim": vecs.shape[0],
                         "extract_descriptors_s": t1 - t0, "compute_score_s": t2 - t1}}
    return metadata, ranks, scores


def rank_images(params, data):
    """Stage ``gandtr_amd.stages.validate.rank_images`` (this build's own name: the reference has no such stage).
    ``params = {"network": ...}``; ``data[0]`` database images, ``data[1]`` query images (omitted: the database queries itself); images are
    tensors, or image FILES (paths / file contents: decoded, resized and normalised on the device, ``data[2] = {"image_size": s}`` optional).
    Returns ``(metadata, ranks, scores)`` -- two output columns after the metadata, the general stage ABI
    (mdir/examples/perform_scenario.py:129)."""
    assert params.keys() == {"network"}, params.keys()
    return _rank_images(params, data)


def validate(params, data):
    """Stage ``mdir.stages.validate.validate`` with the reference's contract (mdir/stages/validate.py:15-39): ``params`` has exactly
    the keys ``network, validation, data`` and the return value is the 1-tuple ``({"eval": {...}},)``.

    The reference builds its validation tasks (datasets from files, mAP / loss criteria) from ``params["validation"]`` and
    ``params["data"]`` (mdir/learning/validation.py): datasets, ground-truth files and ``compute_map`` are out of scope here
    (SURVEY.md section 2), so any non-empty content of those two keys raises ``NotImplementedError`` instead of being ignored.
    With both empty the stage runs the retrieval arithmetic the evaluation is made of on in-memory images (``data[0]`` database,
    ``data[1]`` queries): descriptor extraction -> scores -> ranks; the arrays are returned inside the metadata under
    ``"retrieval": {"ranks", "scores"}``.  ``rank_images`` is the same computation with the arrays as output columns."""
    assert params.keys() == {"network", "validation", "data"}, params.keys()
    for key in ("validation", "data"):
        if params[key]:
            raise NotImplementedError("validate: params[%r] = %r asks for the reference's dataset / criterion machinery "
                                      "(mdir/learning/validation.py), which this build does not provide; pass in-memory images "
                                      "in `data` and leave it empty, or use gandtr_amd.stages.validate.rank_images"
                                      % (key, params[key]))
    metadata, ranks, scores = _rank_images(params, data)
    metadata["retrieval"] = {"ranks": ranks, "scores": scores}
    return (metadata,)
