"""Mixin that lets an nn.Module mirror of a reference model run its forward on the HIP graph executor.

The module keeps ordinary torch Parameters (state_dict compatibility, ``.to(device)``, ``load_state_dict``); when they
live on a cuda (HIP) device the forward is served by a ``gandtr_amd.engine.HipNet`` built from the current
state dict.  The HIP net is rebuilt whenever a parameter/buffer was modified (version counters) or moved.
A cuda device with a missing libgandtr_hip.so raises -- there is no torch fallback on the GPU.
"""
import os

import torch


def default_precision():
    """'f16' (single fp16 MFMA pass, the throughput mode) unless GANDTR_HIP_PRECISION=f16x3 selects the split-fp16 mode that
    reproduces the reference's fp32 results to 1e-3 at every layer."""
    return os.environ.get("GANDTR_HIP_PRECISION", "f16")


class HipBacked:
    #: per-module override of the conv arithmetic on the HIP path: None -> default_precision()
    hip_precision = None

    def _hip_precision(self):
        return self.hip_precision or default_precision()

    def _hip_device(self):
        p = next(self.parameters(), None)
        if p is None:
            p = next(self.buffers(), None)
        return p.device if p is not None else torch.device("cpu")

    def _hip_stamp(self):
        ts = list(self.parameters()) + list(self.buffers())
        return (tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts))

    def _hip_net(self, key, builder):
        """Return the cached HipNet for ``key`` or build it with ``builder(state_dict_on_cpu, device)``."""
        cache = self.__dict__.setdefault("_hip_cache", {})
        stamp = self._hip_stamp()
        if cache.get("__stamp__") != stamp:
            cache.clear()
            cache["__stamp__"] = stamp
        if key not in cache:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            cache[key] = builder(sd, self._hip_device())
        return cache[key]


class ScaledInput:
    """An image batch plus the F.interpolate scale_factor the consumer has to apply (CirMultiscaleAggregation,
    mdir/components/data/wrapper.py:225).  On the HIP path the bilinear resize is fused into the input pack kernel, so
    the pyramid level is never materialised in fp32."""

    def __init__(self, tensor, scale):
        self.tensor, self.scale = tensor, scale

    def to(self, device):
        return ScaledInput(self.tensor.to(device), self.scale)
