"""Data-parallel sharding of the hot path across the GPUs of one node (one process per GPU, torch.distributed:
backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed code at all.  The path shards by independent images: every
rank runs the full network on a contiguous chunk of the batch (weights replicated), and the only exchange step is one
all-gather of the descriptor block -- 2-8 KiB per image, latency-bound, one hop on the fully connected xGMI mesh.
Correctness oracle: the gathered result equals the single-process result on the whole batch.
"""
import torch
import torch.distributed as dist


def chunk_bounds(n_total, world_size, rank):
    """Contiguous chunk [lo, hi) of rank ``rank``; chunks are ceil(n/world) long, the tail may be short or empty."""
    chunk = (n_total + world_size - 1) // world_size
    lo = min(n_total, rank * chunk)
    return lo, min(n_total, lo + chunk), chunk


def shard_batch(x, world_size=None, rank=None):
    """This rank's slice of a batch tensor (first dimension)."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi, _ = chunk_bounds(x.shape[0], world_size, rank)
    return x[lo:hi]


def all_gather_descriptors(local, n_total, group=None):
    """local: [n_local][D] descriptors of this rank's chunk (row-major, the memory layout of the reference's D x N
    view transposed).  Returns the D x n_total matrix (one column per image, imageretrievalnet.py:123) on every rank.
    Ragged batches: the last chunks are zero-padded for the collective and trimmed afterwards."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    _, _, chunk = chunk_bounds(n_total, world, rank)
    d = local.shape[1]
    send = local
    if local.shape[0] != chunk:
        send = torch.zeros((chunk, d), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    out = torch.empty((world * chunk, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    return out[:n_total].t()


def embed_sharded(embed_fn, x, group=None):
    """Run ``embed_fn`` (images -> D x n descriptors) on this rank's chunk of ``x`` and all-gather: D x N on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi, _ = chunk_bounds(x.shape[0], world, rank)
    if hi > lo:
        local = embed_fn(x[lo:hi])
        local = local.reshape(-1, hi - lo).t().contiguous()       # (D,) for a single image (the hub's whitening wrappers squeeze), D x n otherwise
    else:
        local = None
    # ranks with an empty chunk learn D from the others.  Whether ANY rank's chunk is empty follows from (N, world) alone, so every rank knows without asking
    # whether this extra collective is needed: the usual case (N >= world) runs exactly one exchange step, the all-gather
    _, _, chunk = chunk_bounds(x.shape[0], world, rank)
    if (world - 1) * chunk >= x.shape[0]:
        d = torch.tensor([0 if local is None else local.shape[1]], device=x.device if x.is_cuda else "cpu")
        dist.all_reduce(d, op=dist.ReduceOp.MAX, group=group)
        if local is None:
            local = torch.zeros((0, int(d.item())), dtype=torch.float32, device=x.device if x.is_cuda else "cpu")
    return all_gather_descriptors(local, x.shape[0], group)


def descriptors_in_chunks(embed_fn, x, chunk):
    """Single-process counterpart of ``embed_sharded``: the same contiguous chunks of ``chunk`` images, one after the other, concatenated
    to D x N.  Per-chunk shapes (hence kernel / library choices) are those of the sharded run, so the two results are equal bit for bit:
    the multi-GPU correctness contract of SURVEY.md section 8e."""
    cols = []
    for lo in range(0, x.shape[0], chunk):
        hi = min(x.shape[0], lo + chunk)
        cols.append(embed_fn(x[lo:hi]).reshape(-1, hi - lo))
    return torch.cat(cols, dim=1)
