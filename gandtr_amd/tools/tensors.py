"""Nested-container tensor helpers used by the wrapper / network plumbing.
Interface of mdir/tools/tensors.py:8-20 (to_device), :38-64 (MetadataTensor), :67-85 (as_metadata_tensor, as_tensor)."""
from typing import Any, NamedTuple

import torch


def _map(obj, fn, attr):
    if obj is None:
        return None
    if hasattr(obj, attr):
        return fn(obj)
    if isinstance(obj, list):
        return [_map(o, fn, attr) for o in obj]
    if isinstance(obj, tuple):
        return tuple(_map(o, fn, attr) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn, attr) for k, v in obj.items()}
    return fn(obj)


def to_device(tensor, device):
    """Move every tensor of a nested list / tuple / dict structure to ``device`` (structure preserved)."""
    return _map(tensor, lambda t: t.to(device), "to")


def detach(tensor):
    return _map(tensor, lambda t: t.detach(), "detach")


class MetadataTensor(NamedTuple):
    """(tensor, metadata) pair that still collates as a tuple and forwards ``to`` / ``unsqueeze_``."""
    tensor: Any
    metadata: Any

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        meta, = (a.metadata for a in args if isinstance(a, cls))
        plain = [a.tensor if isinstance(a, cls) else a for a in args]
        return MetadataTensor(func(*plain, **(kwargs or {})), meta)

    def __getattr__(self, name):
        attr = getattr(self.tensor, name)
        if name not in ("to", "unsqueeze_"):
            return attr
        return lambda *a, **k: type(self)(attr(*a, **k), self.metadata)


def as_metadata_tensor(tensor, metadata):
    if isinstance(tensor, MetadataTensor):
        tensor.metadata.update(dict(metadata))
        return tensor
    return MetadataTensor(torch.as_tensor(tensor), dict(metadata))


def as_tensor(tensor):
    """Strip MetadataTensor wrappers from a nested structure."""
    if tensor is None:
        return None
    if isinstance(tensor, MetadataTensor):
        return tensor.tensor
    if isinstance(tensor, list):
        return [as_tensor(t) for t in tensor]
    if isinstance(tensor, tuple):
        return tuple(as_tensor(t) for t in tensor)
    if isinstance(tensor, dict):
        return {k: as_tensor(v) for k, v in tensor.items()}
    return torch.as_tensor(tensor)
