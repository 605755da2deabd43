"""Network containers -- host mirror of mdir/learning/network.py: Network :20-95, SingleNetwork :101-292,
SequentialNetwork :635-744, CirSequentialNetwork :747-753, NETWORKS :907-918, initialize_network :920-926.
Only the inference call path is mirrored (training topologies are out of scope)."""
import abc
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


def _plain(x):
    return x if isinstance(x, ScaledInput) else tensors.as_tensor(x)


class Network(abc.ABC):
    TRAIN = "train"
    EVAL = "eval"

    def __init__(self, frozen, model=None):
        self.stage = None
        self.frozen = frozen
        self.model = model
        if frozen:
            self.eval()

    def __call__(self, image):
        return self.forward(image)

    @abc.abstractmethod
    def forward(self, image):
        ...

    @staticmethod
    def initialize_wrappers(wrappers, device):
        if isinstance(wrappers, dict):
            assert wrappers.keys() == {"train", "eval"}, wrappers.keys()
            return {x: initialize_wrappers(wrappers[x], device) for x in wrappers}
        return {x: initialize_wrappers(wrappers, device) for x in ["train", "eval"]}

    def train(self):
        if not self.frozen:
            self.model.train()
            self.stage = Network.TRAIN
        return self

    def eval(self):
        self.model.eval()
        self.stage = Network.EVAL
        return self

    def freeze(self, net="net"):
        assert net == "net"
        self.frozen = True
        self.eval()
        return self

    def parameters(self, optimizer_opts, net="net"):
        assert net == "net"
        if self.frozen:
            return []
        if hasattr(self.model, "parameter_groups"):
            return self.model.parameter_groups(optimizer_opts)
        return self.model.parameters()

    def set_meta(self, meta):
        self.meta = meta
        if self.model:
            self.model.meta = meta


class SingleNetwork(Network):
    """model + stage-dependent wrappers + params (``network_params.model`` / ``network_params.runtime``)."""

    NetworkParams = namedtuple("NetworkParams", ["model", "runtime"])

    def __init__(self, model, network_params, device, frozen):
        self.meta = model.meta if model.meta else {}
        if "model" in network_params.runtime:
            model.runtime = network_params.runtime["model"]
        self.network_params = network_params
        self.wrappers = self.initialize_wrappers(network_params.runtime.get("wrappers", ""), device)
        super().__init__(network_params.runtime.get("frozen", False) or frozen, model.to(device))
        self.device = device
        extra = network_params.runtime.keys() - {"data", "wrappers", "frozen", "model"}
        assert not extra, extra
        extra = network_params.runtime.get("data", {}).keys() - {"mean_std", "transforms"}
        assert not extra, extra

    def forward(self, image, **params):
        return self.wrappers[self.stage](image, self.forward_batch, outputmodel=self.model, tensor_params=params)

    def forward_batch(self, images, **params):
        if images is None:
            return None
        if isinstance(images, list):
            return [self.model(_plain(x), **params) if x is not None else None for x in images]
        return self.model(_plain(images), **params)

    @classmethod
    def initialize(cls, params, device):
        path = params.pop("path", None)
        if not path:
            network_params = cls.NetworkParams(params.pop("model"), params.pop("runtime"))
            model = model_registry.initialize_model(copy.deepcopy(network_params.model))
            init = params.pop("initialize")
            if init and isinstance(init, str):
                with fs_open(init) as handle:
                    model.load_state_dict(torch.load(handle))
            elif init and init["weights"] != "default":
                weights, seed = init.pop("weights"), init.pop("seed")
                torch.manual_seed(seed if seed is not None else time.time())
                model.apply(weight_initialization.initialize_weights(weights, init))
        else:
            with fs_open(path) as handle:
                checkpoint = torch.load(handle, map_location="cpu")
            runtime = params.pop("runtime")
            ck_runtime = checkpoint["network_params"]["runtime"]
            if runtime == "load_from_checkpoint":
                runtime = ck_runtime
            else:
                runtime = {k: (v if v != "load_from_checkpoint" else ck_runtime[k]) for k, v in runtime.items()}
            network_params = cls.NetworkParams(checkpoint["network_params"]["model"], runtime)
            model = model_registry.initialize_model(copy.deepcopy(network_params.model))
            model.load_state_dict(checkpoint["model_state"])
            params.pop("initialize", None)
            if "model" in params:
                pm = params.pop("model")
                assert pm == checkpoint["network_params"]["model"], "{} != {}".format(pm, checkpoint["network_params"]["model"])
        assert not params, params.keys()
        return cls(model, network_params, device=device, frozen=False)

    def overlay_params(self, new_params, device=None):
        if not new_params:
            return self
        new_params["runtime"]["frozen"] = True
        network_params = self.NetworkParams(self.network_params.model, new_params.pop("runtime"))
        assert not new_params
        return self.__class__(self.model, network_params, device or self.device, frozen=True)

    def overlay_model(self, new_model, device=None):
        return self.__class__(new_model, self.network_params, device or self.device, frozen=True)

    def state_dict(self):
        return {"net": {"type": self.__class__.__name__, "frozen": self.frozen,
                        "network_params": self.network_params._asdict(), "model_state": self.model.state_dict()}}

    @classmethod
    def initialize_from_state(cls, state_dict, device, params, runtime):
        assert state_dict.keys() == {"net"}, state_dict.keys()
        checkpoint = state_dict["net"]
        assert checkpoint.keys() == {"type", "frozen", "network_params", "model_state"}, checkpoint.keys()
        network_params = cls.NetworkParams(**checkpoint["network_params"])
        assert checkpoint["type"] == cls.__name__, checkpoint["type"]
        if params is not None and "path" not in params:
            del params["initialize"]
            assert network_params._asdict() == params, "%s != %s" % (network_params._asdict(), params)
        model = model_registry.initialize_model(copy.deepcopy(network_params.model))
        model.load_state_dict(checkpoint["model_state"])
        if runtime:
            network_params.runtime.update(runtime)
        return cls(model, network_params, device=device, frozen=checkpoint["frozen"])

    def __repr__(self):
        params = "\n" + "".join("    %s: %s,\n" % kv for kv in self.network_params._asdict().items())
        wraps = "\n" + "".join("    %s: %s,\n" % (k, indent(str(v))) for k, v in self.wrappers.items())
        return ("%s (\n    meta: %s\n    model: %s\n    network_params: {%s}\n    wrappers: {%s}\n)"
                % (type(self).__name__, self.meta, indent(str(self.model)), indent(params), indent(wraps)))


class SequentialNetwork(Network):
    """Two frozen-or-not networks chained: ``image = first(image); image = last(image)``; the last network's wrappers
    are hoisted to the container (mdir/learning/network.py:639-677)."""

    NetworkParams = namedtuple("NetworkParams", ["runtime"])

    def __init__(self, networks, sequence, device, frozen, rearrange_wrappers=True):
        assert len(networks) == 2
        self.networks = networks
        self.network_order = list(sequence)
        first_net = networks[sequence[0]]
        self.last_net = networks[sequence[1]]
        self.stage = None
        self.frozen = frozen
        self.model = self.last_net.model
        self.device = device
        if rearrange_wrappers:
            self.wrappers = self.last_net.wrappers
            self.last_net.wrappers = self.initialize_wrappers("", device)
            self.network_params = self.NetworkParams({"wrappers": self.last_net.network_params.runtime["wrappers"],
                                                      "data": first_net.network_params.runtime["data"]})
        else:
            self.wrappers = self.initialize_wrappers("", device)
            self.network_params = self.NetworkParams({"wrappers": "", "data": first_net.network_params.runtime["data"]})
        assert first_net.meta["out_channels"] == self.last_net.meta["in_channels"]
        self.meta = {"in_channels": first_net.meta["in_channels"], "out_channels": self.last_net.meta["out_channels"]}
        if frozen:
            self.eval()

    def train(self):
        for n in self.networks.values():
            n.train()
        self.stage = Network.TRAIN
        return self

    def eval(self):
        for n in self.networks.values():
            n.eval()
        self.stage = Network.EVAL
        return self

    def forward(self, image):
        return self.wrappers[self.stage](image, self.forward_batch, outputmodel=self.model)

    def forward_batch(self, images):
        if images is None:
            return None
        if isinstance(images, list):
            return [self._forward_all(x) for x in images]
        return self._forward_all(images)

    def _forward_all(self, image):
        for net in self.network_order:
            image = self.networks[net](image)
        return image

    @classmethod
    def initialize(cls, params, device):
        sequence = params.pop("sequence").split(",")
        rearrange = params.pop("rearrange_wrappers") if "rearrange_wrappers" in params else True
        networks = {name: initialize_network(params.pop(name), device) for name in sequence}
        assert not params, params.keys()
        return cls(networks, sequence, device=device, frozen=False, rearrange_wrappers=rearrange)


class CirSequentialNetwork(SequentialNetwork):
    """Does not split a list of images into single forwards (mdir/learning/network.py:747-753)."""

    def forward_batch(self, images):
        if images is None:
            return None
        return self._forward_all(images)


NETWORKS = {
    "SingleNetwork": SingleNetwork,
    "SequentialNetwork": SequentialNetwork,
    "CirSequentialNetwork": CirSequentialNetwork,
}


def initialize_network(params, device, state=None, runtime=None):
    network_cls = NETWORKS[params.pop("type") if params else state["net"]["type"]]
    if state:
        return network_cls.initialize_from_state(state, device, params, runtime)
    return network_cls.initialize(params, device)
