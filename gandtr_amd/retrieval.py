"""Retrieval scoring on the device -- the consumer of the gathered descriptors.

Reference (CPU numpy): ``scores = np.dot(vecs.T, qvecs); ranks = np.argsort(-scores, axis=0)``
(mdir/components/optim/score/cirscore.py:71-73); hard-negative mining does the same with torch.mm / torch.sort
(mdir/external/cirtorch/datasets/traindataset.py:246-279).  Same layout conventions here: descriptors are D x N (one column
per image, views of the library's [N][D] blocks), scores are Ndb x Nq, ranks are Ndb x Nq database indices per query column.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _hip


def _rows(m):
    """D x N (any strides) -> contiguous [N][D] fp32"""
    return m.t().contiguous().float()


def scores_and_ranks(vecs, qvecs, with_ranks=True, index_base=0):
    """vecs: D x Ndb, qvecs: D x Nq (cuda).  Returns (scores Ndb x Nq fp32, ranks Ndb x Nq int32 or None)."""
    lib = _hip.load()
    if not vecs.is_cuda or vecs.device != qvecs.device:
        raise ValueError("scores_and_ranks needs descriptors on one HIP device")
    v, q = _rows(vecs), _rows(qvecs)
    ndb, d = v.shape
    nq = q.shape[0]
    if q.shape[1] != d:
        raise ValueError("descriptor sizes differ: %d vs %d" % (d, q.shape[1]))
    need = ctypes.c_size_t()
    with torch.cuda.device(v.device):
        _hip.check(lib.gdt_retrieval_workspace_bytes(ndb, nq, d, int(with_ranks), ctypes.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=v.device)
        scores_t = torch.empty((nq, ndb), dtype=torch.float32, device=v.device)
        ranks_t = torch.empty((nq, ndb), dtype=torch.int32, device=v.device) if with_ranks else None
        _hip.check(lib.gdt_retrieval_scores_ranks(v.data_ptr(), q.data_ptr(), scores_t.data_ptr(),
                                                  ranks_t.data_ptr() if with_ranks else None, ndb, nq, d, index_base,
                                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream(v.device).cuda_stream))
    return scores_t.t(), (ranks_t.t() if with_ranks else None)


def select_negatives(ranks, pool_clusters, query_clusters, vecs, qvecs, nnum, index_base=0):
    """The selection loop of the reference's hard-negative mining on the device (traindataset.py:256-275): per query (column of ``ranks``,
    Ndb x Nq as returned by ``scores_and_ranks``) the first ``nnum`` pool positions whose cluster is neither the query's nor that of a
    position already taken.  pool_clusters: Ndb ints, query_clusters: Nq ints.  Returns (positions Nq x nnum int32, distances Nq x nnum fp32 =
    ||q - p + 1e-6||_2, the reference's statistic).  Raises IndexError where the reference's ``ranks[r, q]`` would run past the pool."""
    lib = _hip.load()
    dev = vecs.device
    v, q = _rows(vecs), _rows(qvecs)
    ndb, d = v.shape
    nq = q.shape[0]
    rk = ranks.t().contiguous().to(torch.int32)                               # [Nq][Ndb]
    pc = torch.as_tensor(pool_clusters, dtype=torch.int32, device=dev).contiguous()
    qc = torch.as_tensor(query_clusters, dtype=torch.int32, device=dev).contiguous()
    if rk.shape != (nq, ndb) or pc.numel() != ndb or qc.numel() != nq:
        raise ValueError("ranks is Ndb x Nq, pool_clusters has Ndb and query_clusters Nq entries")
    pos = torch.empty((nq, nnum), dtype=torch.int32, device=dev)
    dist_ = torch.empty((nq, nnum), dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(lib.gdt_retrieval_select_negatives(rk.data_ptr(), pc.data_ptr(), qc.data_ptr(), v.data_ptr(), q.data_ptr(), pos.data_ptr(),
                                                      dist_.data_ptr(), status.data_ptr(), ndb, nq, d, int(nnum), int(index_base),
                                                      torch.cuda.current_stream(dev).cuda_stream))
    if int(status.item()) & 1:
        raise IndexError("hard-negative selection ran past the pool: fewer than %d other clusters among its images" % nnum)
    return pos, dist_


def search_hard_negatives(qidxs, qvecs, idxs2images, poolvecs, clusters, nnum):
    """``TuplesDataset._search_hard_negatives`` (traindataset.py:246-279) with scores, sort AND the cluster-aware selection on the device.
    qidxs: image index per query; qvecs: D x Nq; idxs2images: image index per pool position; poolvecs: D x Npool; clusters: cluster id per image
    index (``self.clusters``); nnum negatives per query.  Returns what the reference returns: (nidxs -- a list of nnum image indices per query --,
    {"average_negative_distance": [one l2 distance per chosen negative, query by query]})."""
    clusters = torch.as_tensor(clusters)
    idxs2images_t = torch.as_tensor(idxs2images, dtype=torch.int64)
    _, ranks = scores_and_ranks(poolvecs, qvecs)
    pos, dist_ = select_negatives(ranks, clusters[idxs2images_t], clusters[torch.as_tensor(qidxs, dtype=torch.int64)], poolvecs, qvecs, nnum)
    nidxs = idxs2images_t[pos.cpu().long()].tolist()
    return nidxs, {"average_negative_distance": dist_.cpu().reshape(-1).tolist()}


def sharded_topk(vecs_local, qvecs, k, group=None):
    """Database sharded over the ranks of one node (contiguous chunks, as gandtr_amd.sharding), queries replicated.
    Every rank scores its shard, keeps its local top-k per query and all-gathers the candidates (k scores + k global ids per
    query and rank: tiny, latency-bound); the final order is the top-k of the world*k candidates.
    Returns (scores k x Nq, ids k x Nq) identical on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_local = torch.tensor([vecs_local.shape[1]], device=vecs_local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    base = int(sum(int(c.item()) for c in counts[:rank]))
    nq = qvecs.shape[1]
    kk = min(k, max(int(c.item()) for c in counts))
    cand_s = torch.full((nq, kk), -float("inf"), dtype=torch.float32, device=qvecs.device)
    cand_i = torch.full((nq, kk), -1, dtype=torch.int32, device=qvecs.device)
    if vecs_local.shape[1] > 0:
        scores, ranks = scores_and_ranks(vecs_local, qvecs, True, index_base=base)
        top = min(kk, vecs_local.shape[1])
        ids = ranks.t()[:, :top].contiguous()                                  # nq x top, global ids
        cand_i[:, :top] = ids
        cand_s[:, :top] = torch.gather(scores.t(), 1, (ids - base).long())
    all_s = torch.empty((world * nq, kk), dtype=torch.float32, device=qvecs.device)      # rank blocks stacked along dim 0
    all_i = torch.empty((world * nq, kk), dtype=torch.int32, device=qvecs.device)
    dist.all_gather_into_tensor(all_s, cand_s, group=group)
    dist.all_gather_into_tensor(all_i, cand_i, group=group)
    flat_s = all_s.view(world, nq, kk).permute(1, 0, 2).reshape(nq, world * kk)
    flat_i = all_i.view(world, nq, kk).permute(1, 0, 2).reshape(nq, world * kk)
    order = torch.argsort(flat_s, dim=1, descending=True, stable=True)[:, :k]   # world*k candidates per query: bookkeeping
    return torch.gather(flat_s, 1, order).t(), torch.gather(flat_i, 1, order.long()).t()
