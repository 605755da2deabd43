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
        q.put(out.clone())
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
    got = q.get(timeout=120)
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
