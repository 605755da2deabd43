"""Model registry (plugin API) -- mirror of mdir/components/model/network/__init__.py:20-48 restricted to the
architectures reachable from the hub entrypoints and BASELINE configs.  Unknown names raise KeyError like the
reference (:48)."""
import torch.nn as nn

from . import cirnet, hed, p2p_networks


class Identity(nn.Module):
    def __init__(self):
        super().__init__()
        self.meta = {"out_channels": 3, "in_channels": 3}

    def forward(self, x):
        return x


MODEL_LABELS = {
    "identity": Identity,
    "official_resnet_generator": p2p_networks.ResnetGenerator,
    "cirnet": cirnet.init_cirnet,
    "hed_interpolation": hed.HedInterpolation,
}


def initialize_model(params):
    return MODEL_LABELS[params.pop("architecture")](**params)
