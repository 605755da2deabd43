"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, embed their chunk and all-gather the descriptors;
the result must equal the single-process result on the whole batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gandtr_amd import sharding
from gandtr_amd.tools import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _tiny_embed(x):
    """stand-in embedder (D x n): deterministic function of each image alone"""
    w = synth._normal(0, "proj", (12, 16))
    f = torch.nn.functional.adaptive_avg_pool2d(x, 2).flatten(1)          # n x 12
    v = f @ w
    return (v / v.norm(dim=1, keepdim=True)).t()


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    x = synth.synth_input(11, (n_total, 3, 8, 8))
    lo, hi, chunk = sharding.chunk_bounds(n_total, world, rank)
    assert sharding.shard_batch(x).shape[0] == hi - lo
    out = sharding.embed_sharded(_tiny_embed, x)
    if rank == 0:
        q.put(out.clone().numpy())           # (by value: a torch tensor travels as a shared-memory handle that dies with this process -- a race with the parent's get)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])
def test_sharded_embedding_equals_single_process(n_total):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _tiny_embed(synth.synth_input(11, (n_total, 3, 8, 8)))
    assert got.shape == ref.shape == (16, n_total)
    assert torch.equal(got, ref)


def test_chunk_bounds_cover_batch():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [sharding.chunk_bounds(n, w, r)[:2] for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _np_scores_and_ranks(vecs, qvecs, with_ranks=True, index_base=0):
    """numpy statement of the reference's scoring (cirscore.py:71-73), standing in for the HIP kernel on CPU ranks"""
    import numpy as np
    s = np.dot(vecs.numpy().T, qvecs.numpy())
    r = np.argsort(-s, axis=0, kind="stable").astype(np.int32) + index_base
    return torch.from_numpy(s), torch.from_numpy(r)


def _topk_worker(rank, world, port, ndb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from gandtr_amd import retrieval
    retrieval.scores_and_ranks = _np_scores_and_ranks
    vecs = synth._normal(5, "db", (32, ndb))
    qv = synth._normal(6, "q", (32, 4))
    lo, hi, _ = sharding.chunk_bounds(ndb, world, rank)
    s, i = retrieval.sharded_topk(vecs[:, lo:hi], qv, k=5)
    if rank == 0:
        q.put((s.clone().numpy(), i.clone().numpy()))       # (by value, see above)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ndb", [40, 7, 3])
def test_sharded_topk_equals_global_ranking(ndb):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_topk_worker, args=(r, 2, port, ndb, q)) for r in range(2)]
    for p in procs:
        p.start()
    s, i = (torch.from_numpy(a) for a in q.get(timeout=120))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    vecs, qv = synth._normal(5, "db", (32, ndb)), synth._normal(6, "q", (32, 4))
    ref_s, ref_r = _np_scores_and_ranks(vecs, qv)
    k = min(5, ndb)
    assert torch.equal(i[:k].long(), ref_r[:k].long())
    assert torch.allclose(s[:k], torch.gather(ref_s, 0, ref_r[:k].long()))
