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
