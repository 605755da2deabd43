"""Mixin that lets an nn.Module mirror of a reference model run its forward on the HIP graph executor.

The module keeps ordinary torch Parameters (state_dict compatibility, ``.to(device)``, ``load_state_dict``); when they
live on a cuda (HIP) device the forward is served by a ``gandtr_amd.engine.HipNet`` built from the current
state dict.  The HIP net is rebuilt whenever a parameter/buffer was modified (version counters) or moved.
A cuda device with a missing libgandtr_hip.so raises -- there is no torch fallback on the GPU.
"""
import os

import torch


def default_precision(model_default="f16"):
    """Conv arithmetic of the HIP path when the module does not say: GANDTR_HIP_PRECISION if set, else the model family's own
    default -- the mode in which that family meets north_star's parity gates: "f16c" for the generators
    (single-pass fp16 leaves their 24-layer stack at 2.5e-3 of the fp32 reference), "f16" for the embedders and HED."""
    return os.environ.get("GANDTR_HIP_PRECISION") or model_default


class HipBacked:
    #: per-module override of the conv arithmetic on the HIP path ("f16" | "f16c" | "f16x3"): None -> default_precision()
    hip_precision = None
    #: the family default (see default_precision)
    hip_default_precision = "f16"

    def _hip_precision(self):
        return self.hip_precision or default_precision(self.hip_default_precision)

    def _hip_check_inference(self):
        """The HIP forward returns tensors without autograd history.  A module in training mode with trainable parameters and
        autograd enabled expects gradients: refuse.  In eval mode (how the hub returns its networks, mdir/hub/model.py:34-36) the
        call is served, with one warning that no graph is recorded unless the caller used torch.no_grad()."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if self.training:
                raise NotImplementedError("the HIP path is inference-only: call .eval() and wrap the call in torch.no_grad(); "
                                          "training runs on the torch modules (device='cpu')")
            if not HipBacked._warned_no_grad:
                HipBacked._warned_no_grad = True
                import warnings
                warnings.warn("gandtr_amd: the HIP forward records no autograd graph; wrap inference in torch.no_grad()", stacklevel=3)

    _warned_no_grad = False

    def _hip_device(self):
        ts = self._hip_tensors()
        return ts[0].device if ts else torch.device("cpu")

    def _hip_tensors(self):
        """the module tree's parameters and buffers, listed once (walking 625 tensors of a ResNet-101 through nn.Module's generators took 0.8 of the 0.94 ms a
        stamp cost per call); dropped whenever the module tree may hold other tensor objects (``_apply``: .to / .cuda / .half; ``load_state_dict``)"""
        ts = self.__dict__.get("_hip_ts")
        if ts is None:
            ts = list(self.parameters()) + list(self.buffers())
            self.__dict__["_hip_ts"] = ts
        return ts

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_hip_ts", None)
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_hip_ts", None)
        return out

    def load_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_hip_ts", None)
        out = super().load_state_dict(*args, **kwargs)
        self.__dict__.pop("_hip_ts", None)
        return out

    def _hip_stamp(self):
        ts = self._hip_tensors()
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
