"""Inference containers of the hot path, written against the drop-in contract of SURVEY.md section 8b (the reference interface is
mdir/learning/network.py: Network :20-95, SingleNetwork :101-292, SequentialNetwork :635-744, CirSequentialNetwork :747-753,
NETWORKS :907-918, initialize_network :920-926).

What the contract fixes, and nothing more is kept:
  * construction from a params dict (``type``, ``model``, ``runtime``, ``initialize`` | ``path``) or from a saved state
    (``{"net": {"type", "frozen", "network_params", "model_state"}}``), through the registries ``NETWORKS`` / ``MODEL_LABELS`` /
    ``WRAPPERS_LABELS``;
  * the attribute surface the hub, the stages and user code touch: ``model, wrappers, network_params, meta, stage, device, frozen``
    (+ ``transform`` attached by the hub), ``__call__/forward/forward_batch/eval/train/freeze/parameters/state_dict/overlay_*``;
  * the call path: stage-dependent wrapper chain around ``model(x)``; a list input is a Python loop of forwards; the augment ->
    embed chain feeds one network's output to the next and hoists the last network's wrappers to the container;
  * the reference's error behaviour: ``AssertionError`` for unknown runtime keys / left-over params, ``KeyError`` for unknown types.
Training topologies (network sets, multi-head, global-local) are out of scope.
"""
import copy
import time
from collections import namedtuple

import torch

from ..components.data.wrapper import initialize_wrappers
from ..components.model import network as model_registry
from ..components.model import weight_initialization
from ..components.model.network._hipbacked import ScaledInput
from ..tools import tensors
from ..tools.utils import fs_open, indent

TRAIN, EVAL = "train", "eval"
_RUNTIME_KEYS = {"data", "wrappers", "frozen", "model"}
_DATA_KEYS = {"mean_std", "transforms"}


def _wrapper_chains(spec, device):
    """``{"train": Compose, "eval": Compose}`` from one spec for both stages or a dict with exactly those two keys."""
    if isinstance(spec, dict):
        assert spec.keys() == {TRAIN, EVAL}, spec.keys()
        return {stage: initialize_wrappers(spec[stage], device) for stage in (TRAIN, EVAL)}
    return {stage: initialize_wrappers(spec, device) for stage in (TRAIN, EVAL)}


def _as_model_input(x):
    return x if isinstance(x, ScaledInput) else tensors.as_tensor(x)


def load_restricted(path):
    """torch.load with the restricted unpickler; a checkpoint that carries a type outside torch's allow-list fails with the name of that
    type and the way to admit it, instead of a bare UnpicklingError from inside hubconf.*(pretrained=True)."""
    import pickle
    with fs_open(str(path)) as handle:
        try:
            return torch.load(handle, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as exc:
            raise RuntimeError(
                f"checkpoint {path} holds an object the restricted loader does not admit ({exc}). If the type named above is benign, "
                "register it with torch.serialization.add_safe_globals([...]) before loading; unrestricted unpickling of downloaded "
                "files is deliberately not offered.") from exc


def _load_torch(path):
    """checkpoint payloads are dicts / lists / strings / tensors: no arbitrary unpickling"""
    return load_restricted(path)


class Network:
    """Stage bookkeeping shared by the containers.  ``stage`` selects the wrapper chain; a frozen network stays in eval mode."""

    TRAIN, EVAL = TRAIN, EVAL

    stage = None
    frozen = False
    model = None

    def __call__(self, image):
        return self.forward(image)

    def forward(self, image):
        raise NotImplementedError

    # kept as a static method: the reference exposes it under this name
    initialize_wrappers = staticmethod(_wrapper_chains)

    def _members(self):
        return (self.model,)

    def train(self):
        if self.frozen:
            return self
        for m in self._members():
            m.train()
        self.stage = TRAIN
        return self

    def eval(self):
        for m in self._members():
            m.eval()
        self.stage = EVAL
        return self

    def freeze(self, net="net"):
        assert net == "net"
        self.frozen = True
        return self.eval()

    def parameters(self, optimizer_opts, net="net"):
        assert net == "net"
        if self.frozen:
            return []
        groups = getattr(self.model, "parameter_groups", None)
        return groups(optimizer_opts) if groups else self.model.parameters()

    def set_meta(self, meta):
        self.meta = meta
        if self.model:
            self.model.meta = meta


class SingleNetwork(Network):
    """One model, its wrapper chains and the params it was built from."""

    NetworkParams = namedtuple("NetworkParams", ["model", "runtime"])

    def __init__(self, model, network_params, device, frozen):
        runtime = network_params.runtime
        unknown = runtime.keys() - _RUNTIME_KEYS
        assert not unknown, unknown
        unknown = runtime.get("data", {}).keys() - _DATA_KEYS
        assert not unknown, unknown
        self.meta = model.meta if model.meta else {}
        if "model" in runtime:
            model.runtime = runtime["model"]
        self.network_params = network_params
        self.device = device
        self.wrappers = _wrapper_chains(runtime.get("wrappers", ""), device)
        self.model = model.to(device)
        self.frozen = bool(runtime.get("frozen", False) or frozen)
        if self.frozen:
            self.eval()

    # ---- call path
    def forward(self, image, **params):
        chain = self.wrappers[self.stage]
        return chain(image, self.forward_batch, outputmodel=self.model, tensor_params=params, fold_input=True)

    def forward_list(self, images):
        """``[self.forward(x) for x in images]`` (the reference's loop over a dataset, imageretrievalnet.py:319-333) with the model forwards of ALL the images --
        every level of every image's pyramid -- handed to the model in one ``forward_many`` call: on a HIP device they are issued on side streams and fill each
        other's gaps (one 1024 x 683 image at three scales leaves most of the chip idle).  Pre- and postprocessing wrappers run per image, in the reference's
        order; falls back to the plain loop when the chain folds input wrappers into the model or the model has no ``forward_many``."""
        chain = self.wrappers[self.stage]
        if len(images) < 2 or not chain.wrappers or not hasattr(self.model, "forward_many"):
            return [self.forward(x) for x in images]
        active, folded = chain._fold_input_wrappers(self.model)
        if folded is not None:
            return [self.forward(x) for x in images]
        pre, flat = [], []
        for x in images:
            metas = []
            for w in active:
                x, meta = w.preprocess(x, self.model)
                metas.append(meta)
            x = tensors.to_device(x, chain.device)
            items = x if isinstance(x, list) else [x]
            if any(i is None for i in items):
                return [self.forward(x) for x in images]
            pre.append((metas, isinstance(x, list), len(items)))
            flat.extend(_as_model_input(i) for i in items)
        outs = self.model.forward_many(flat)
        results, at = [], 0
        for metas, was_list, n in pre:
            y = outs[at:at + n] if was_list else outs[at]
            at += n
            for w, meta in reversed(list(zip(active, metas))):
                y = w.postprocess(y, self.model, meta)
            results.append(y)
        return results

    def forward_batch(self, images, **params):
        if images is None:
            return None
        if isinstance(images, list):
            if not params and len(images) > 1 and hasattr(self.model, "forward_many") and all(x is not None for x in images):
                return self.model.forward_many([_as_model_input(x) for x in images])          # pyramid levels: concurrent on a HIP device
            return [None if x is None else self.model(_as_model_input(x), **params) for x in images]
        return self.model(_as_model_input(images), **params)

    # ---- construction
    @classmethod
    def initialize(cls, params, device):
        path = params.pop("path", None)
        if path:
            model, network_params = cls._from_checkpoint_file(path, params)
        else:
            network_params = cls.NetworkParams(params.pop("model"), params.pop("runtime"))
            model = model_registry.initialize_model(copy.deepcopy(network_params.model))
            cls._initialize_weights(model, params.pop("initialize"))
        assert not params, params.keys()
        return cls(model, network_params, device=device, frozen=False)

    @staticmethod
    def _initialize_weights(model, init):
        """``initialize``: falsy -> the modules' own default init; a path -> a plain state dict; a dict -> seeded registry init."""
        if not init:
            return
        if isinstance(init, str):
            model.load_state_dict(_load_torch(init))
            return
        if init["weights"] == "default":
            return
        scheme, seed = init.pop("weights"), init.pop("seed")
        torch.manual_seed(seed if seed is not None else time.time())
        model.apply(weight_initialization.initialize_weights(scheme, init))

    @classmethod
    def _from_checkpoint_file(cls, path, params):
        checkpoint = _load_torch(path)
        saved = checkpoint["network_params"]
        runtime = params.pop("runtime")
        if runtime == "load_from_checkpoint":
            runtime = saved["runtime"]
        else:
            runtime = {k: (saved["runtime"][k] if v == "load_from_checkpoint" else v) for k, v in runtime.items()}
        params.pop("initialize", None)
        if "model" in params:
            given = params.pop("model")
            assert given == saved["model"], "{} != {}".format(given, saved["model"])
        model = model_registry.initialize_model(copy.deepcopy(saved["model"]))
        model.load_state_dict(checkpoint["model_state"])
        return model, cls.NetworkParams(saved["model"], runtime)

    @classmethod
    def initialize_from_state(cls, state_dict, device, params, runtime):
        assert state_dict.keys() == {"net"}, state_dict.keys()
        saved = state_dict["net"]
        assert saved.keys() == {"type", "frozen", "network_params", "model_state"}, saved.keys()
        assert saved["type"] == cls.__name__, saved["type"]
        network_params = cls.NetworkParams(**saved["network_params"])
        if params is not None and "path" not in params:
            del params["initialize"]
            assert network_params._asdict() == params, "%s != %s" % (network_params._asdict(), params)
        model = model_registry.initialize_model(copy.deepcopy(network_params.model))
        model.load_state_dict(saved["model_state"])
        if runtime:
            network_params.runtime.update(runtime)
        return cls(model, network_params, device=device, frozen=saved["frozen"])

    # ---- (de)serialisation, overlays
    def state_dict(self):
        return {"net": {"type": type(self).__name__, "frozen": self.frozen, "network_params": self.network_params._asdict(),
                        "model_state": self.model.state_dict()}}

    def overlay_params(self, new_params, device=None):
        if not new_params:
            return self
        runtime = new_params.pop("runtime")
        assert not new_params
        runtime["frozen"] = True
        return type(self)(self.model, self.NetworkParams(self.network_params.model, runtime), device or self.device, frozen=True)

    def overlay_model(self, new_model, device=None):
        return type(self)(new_model, self.network_params, device or self.device, frozen=True)

    def __repr__(self):
        params = "\n" + "".join("    %s: %s,\n" % kv for kv in self.network_params._asdict().items())
        wraps = "\n" + "".join("    %s: %s,\n" % (k, indent(str(v))) for k, v in self.wrappers.items())
        return ("%s (\n    meta: %s\n    model: %s\n    network_params: {%s}\n    wrappers: {%s}\n)"
                % (type(self).__name__, self.meta, indent(str(self.model)), indent(params), indent(wraps)))


class SequentialNetwork(Network):
    """Two networks in a row (``sequence="first,last"``): ``last(first(image))``.  By default the last network's wrappers move to
    the container, so that e.g. multi-scale aggregation wraps the whole chain and not just the embedder."""

    NetworkParams = namedtuple("NetworkParams", ["runtime"])

    def __init__(self, networks, sequence, device, frozen, rearrange_wrappers=True):
        assert len(networks) == 2
        self.networks = networks
        self.network_order = list(sequence)
        first, last = networks[sequence[0]], networks[sequence[1]]
        assert first.meta["out_channels"] == last.meta["in_channels"]
        self.last_net = last
        self.model = last.model
        self.device = device
        self.frozen = frozen
        self.meta = {"in_channels": first.meta["in_channels"], "out_channels": last.meta["out_channels"]}
        data = first.network_params.runtime["data"]
        if rearrange_wrappers:
            self.wrappers, last.wrappers = last.wrappers, _wrapper_chains("", device)
            self.network_params = self.NetworkParams({"wrappers": last.network_params.runtime["wrappers"], "data": data})
        else:
            self.wrappers = _wrapper_chains("", device)
            self.network_params = self.NetworkParams({"wrappers": "", "data": data})
        if frozen:
            self.eval()

    def _members(self):
        return tuple(self.networks.values())

    def train(self):                        # member networks apply their own frozen flags
        for n in self.networks.values():
            n.train()
        self.stage = TRAIN
        return self

    def forward(self, image):
        return self.wrappers[self.stage](image, self.forward_batch, outputmodel=self.model)

    def _chain(self, image):
        for name in self.network_order:
            image = self.networks[name](image)
        return image

    def forward_batch(self, images):
        if images is None:
            return None
        if isinstance(images, list):
            return [self._chain(x) for x in images]
        return self._chain(images)

    @classmethod
    def initialize(cls, params, device):
        sequence = params.pop("sequence").split(",")
        rearrange = params.pop("rearrange_wrappers", True)
        networks = {name: initialize_network(params.pop(name), device) for name in sequence}
        assert not params, params.keys()
        return cls(networks, sequence, device=device, frozen=False, rearrange_wrappers=rearrange)


class CirSequentialNetwork(SequentialNetwork):
    """The retrieval flavour hands a list of images to the chain as it is (the embedder's tuple-batch wrapper deals with it)."""

    def forward_batch(self, images):
        return None if images is None else self._chain(images)


NETWORKS = {
    "SingleNetwork": SingleNetwork,
    "SequentialNetwork": SequentialNetwork,
    "CirSequentialNetwork": CirSequentialNetwork,
}


def initialize_network(params, device, state=None, runtime=None):
    kind = params.pop("type") if params else state["net"]["type"]
    cls = NETWORKS[kind]
    if state:
        return cls.initialize_from_state(state, device, params, runtime)
    return cls.initialize(params, device)
