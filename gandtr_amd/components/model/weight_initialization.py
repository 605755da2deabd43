"""Weight initialisers executed by ``SingleNetwork.initialize`` for ``pretrained=False`` hub models
(mdir/learning/network.py:152-162).  Behaviour of mdir/components/model/weight_initialization.py:54-87:
``normal_p2p`` / ``kaiming_p2p`` with default init_gain 0.2 (:82-83); Conv/Linear weights ~ N(0, gain) or
kaiming-normal(fan_in), biases 0; BatchNorm2d weight ~ N(1, gain), bias 0.  Applied with ``model.apply`` after
``torch.manual_seed(seed)`` so that the module traversal order -- identical to the reference's because the
module tree is identical -- reproduces the reference's weights bit for bit."""
import math

import torch.nn as nn


def init_weights_p2p(init_type, init_gain):
    if init_type not in ("normal", "kaiming"):
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)

    def init_func(m):
        cname = type(m).__name__
        if hasattr(m, "weight") and ("Conv" in cname or "Linear" in cname):
            if init_type == "normal":
                nn.init.normal_(m.weight.data, 0.0, init_gain)
            else:
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif "BatchNorm2d" in cname:
            nn.init.normal_(m.weight.data, 1.0, init_gain)
            nn.init.constant_(m.bias.data, 0.0)
    return init_func


def _simple(fn_w, fn_b):
    def init_func(m):
        cname = type(m).__name__
        if hasattr(m, "weight") and ("Conv" in cname or "Linear" in cname):
            fn_w(m.weight.data)
            if getattr(m, "bias", None) is not None:
                fn_b(m.bias.data)
    return init_func


def _he_normal(w):
    fan_in = w.size(1) * (w[0][0].numel() if w.dim() > 2 else 1)
    return w.normal_(0, math.sqrt(2.0 / fan_in))


WEIGHT_INITIALIZATIONS = {
    "uniform": _simple(nn.init.uniform_, nn.init.uniform_),
    "normal": _simple(nn.init.normal_, nn.init.normal_),
    "he_normal": _simple(_he_normal, lambda b: nn.init.constant_(b, 0.01)),
}


def initialize_weights(weights, params):
    if "p2p" in weights:
        if params is None or "init_gain" not in params:
            params = {"init_gain": 0.2}
        return init_weights_p2p(weights.split("_")[0], **params)
    assert not params
    return WEIGHT_INITIALIZATIONS[weights]
