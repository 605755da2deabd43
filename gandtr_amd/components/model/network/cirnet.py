"""GeM retrieval embedder -- host mirror of
  mdir/external/cirtorch/layers/functional.py:21-22 (gem), :130-131 (l2n)
  mdir/external/cirtorch/layers/pooling.py:36-47 (GeM), layers/normalization.py:10-20 (L2N)
  mdir/external/cirtorch/networks/imageretrievalnet.py:86-123 (ImageRetrievalNet), :146-309 (init_network)
  mdir/components/model/network/cirnet.py:8-65 (CirRetrievalNet, init_cirnet)
Only the hub configuration is on the HIP hot path: GeM pooling, no local / regional / final whitening layers
(mdir/hub/embedding.yml:6-10).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import backbones
from ._hipbacked import HipBacked, ScaledInput


def gem(x, p=3, eps=1e-6):
    return F.avg_pool2d(x.clamp(min=eps).pow(p), (x.size(-2), x.size(-1))).pow(1. / p)


def l2n(x, eps=1e-6):
    return x / (torch.norm(x, p=2, dim=1, keepdim=True) + eps).expand_as(x)


class GeM(nn.Module):
    def __init__(self, p=3, eps=1e-6):
        super().__init__()
        self.p = Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x):
        return gem(x, p=self.p, eps=self.eps)

    def __repr__(self):
        return "%s(p=%.4f, eps=%s)" % (type(self).__name__, self.p.data.tolist()[0], self.eps)


class L2N(nn.Module):
    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, x):
        return l2n(x, eps=self.eps)

    def __repr__(self):
        return "%s(eps=%s)" % (type(self).__name__, self.eps)


OUTPUT_DIM = {"vgg16": 512, "resnet50": 2048, "resnet101": 2048, "resnet152": 2048}


class ImageRetrievalNet(HipBacked, nn.Module):
    """features -> GeM -> L2N; returns D x N (one column per image)."""

    def __init__(self, features, lwhiten, pool, whiten, meta):
        super().__init__()
        self.features = features if isinstance(features, nn.Sequential) else nn.Sequential(*features)
        self.lwhiten = lwhiten
        self.pool = pool
        self.whiten = whiten
        self.norm = L2N()
        self.meta = meta

    def forward(self, x):
        scale = None
        if isinstance(x, ScaledInput):
            x, scale = x.tensor, x.scale
        if self._hip_device().type == "cuda":
            return self._forward_hip(x, scale)
        if scale is not None:
            x = F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
        o = self.features(x)
        if self.lwhiten is not None:
            s = o.size()
            o = self.lwhiten(o.permute(0, 2, 3, 1).contiguous().view(-1, s[1]))
            o = o.view(s[0], s[2], s[3], self.lwhiten.out_features).permute(0, 3, 1, 2)
        o = self.norm(self.pool(o)).squeeze(-1).squeeze(-1)
        if self.whiten is not None:
            o = self.norm(self.whiten(o))
        return o.permute(1, 0)

    def _forward_hip(self, x, scale):
        from .... import engine
        if self.lwhiten is not None or self.whiten is not None or not isinstance(self.pool, GeM):
            raise NotImplementedError("HIP embedder supports GeM pooling without local/final whitening layers (hub configuration)")
        if self.meta.get("architecture") not in ("vgg16", "resnet50", "resnet101", "resnet152"):
            raise NotImplementedError("HIP embedder supports vgg16 / resnet50 / resnet101 / resnet152 trunks")
        self._hip_check_inference()
        prec = self._hip_precision()
        net = self._hip_net(("embed", prec), lambda sd, dev: engine.build_embedder(sd, dev, precision=prec))
        return net.forward(x, scale=scale)[net.out_slot].t()      # N x D storage, D x N view (imageretrievalnet.py:123)

    def forward_many(self, xs):
        """``[self(x) for x in xs]`` with the HIP forwards of the list issued concurrently (engine.HipNet.forward_many): the levels of a
        multi-scale pyramid are independent and each is too small to fill the chip."""
        from .... import engine
        if self._hip_device().type != "cuda" or len(xs) < 2:
            return [self(x) for x in xs]
        if self.lwhiten is not None or self.whiten is not None or not isinstance(self.pool, GeM) or \
                self.meta.get("architecture") not in ("vgg16", "resnet50", "resnet101", "resnet152"):
            return [self(x) for x in xs]                          # (raises the same NotImplementedError as the single call)
        self._hip_check_inference()
        prec = self._hip_precision()
        net = self._hip_net(("embed", prec), lambda sd, dev: engine.build_embedder(sd, dev, precision=prec))
        pairs = [(x.tensor, x.scale) if isinstance(x, ScaledInput) else (x, None) for x in xs]
        return [outs[net.out_slot].t() for outs in net.forward_many(pairs)]

    def meta_repr(self):
        lines = ["  (meta): dict("]
        for k in ("architecture", "local_whitening", "pooling", "regional", "whitening", "outputdim", "mean", "std"):
            lines.append("     %s: %s" % (k, self.meta.get(k)))
        return "\n".join(lines) + "\n  )\n"

    def __repr__(self):
        return super().__repr__()[:-1] + self.meta_repr() + ")"


def init_network(params):
    """cirtorch init_network reduced to the configurations reachable from the hub / cirnet registry entry:
    random-initialised trunk (weights arrive through load_state_dict), GeM pooling, optional plain-Linear whitening
    layers are rejected on the hot path."""
    architecture = params.get("architecture", "resnet101")
    local_whitening = params.get("local_whitening", False)
    pooling = params.get("pooling", "gem")
    regional = params.get("regional", False)
    whitening = params.get("whitening", False)
    mean = params.get("mean", [0.485, 0.456, 0.406])
    std = params.get("std", [0.229, 0.224, 0.225])
    pretrained = params.get("pretrained", True)

    if architecture not in backbones.ARCHITECTURES:
        raise ValueError("Unsupported or unknown architecture: {}!".format(architecture))
    if pretrained:
        raise ValueError("pretrained ImageNet trunks need a download; use pretrained=False and load a checkpoint")
    net_in = backbones.ARCHITECTURES[architecture](pretrained=False)
    if architecture.startswith("vgg"):
        features = list(net_in.features.children())[:-1]      # drop the last MaxPool
    else:
        features = list(net_in.children())[:-2]               # drop avgpool, fc
    last_convs = [m for f in features[-2:] for m in f.modules() if isinstance(m, nn.Conv2d)]
    dim = last_convs[-1].out_channels
    if local_whitening or regional or whitening:
        raise NotImplementedError("local / regional / final whitening layers are outside the gandtr hot path "
                                  "(mdir/hub/embedding.yml uses none)")
    if pooling != "gem":
        raise NotImplementedError("only GeM pooling is on the gandtr hot path")
    meta = {"architecture": architecture, "local_whitening": local_whitening, "pooling": pooling, "regional": regional,
            "whitening": whitening, "mean": mean, "std": std, "outputdim": dim, "out_channels": dim}
    return ImageRetrievalNet(features, None, GeM(), None, meta)


class CirRetrievalNet(ImageRetrievalNet):
    """cirtorch retrieval net with the optimiser parameter groups of the reference (pool exponent: 10x lr, no weight
    decay) and BatchNorm layers frozen in eval mode while training."""

    def parameter_groups(self, optimizer_opts):
        return [{"params": self.features.parameters()},
                {"params": self.pool.parameters(), "lr": optimizer_opts["lr"] * 10, "weight_decay": 0}]

    def train(self, mode=True):
        res = super().train(mode)
        if mode:
            for m in self.modules():
                if "BatchNorm" in type(m).__name__:
                    m.eval()
        return res


def init_cirnet(**params):
    for key in ("local_whitening", "pooling", "regional", "whitening", "pretrained"):
        if key not in params:
            raise ValueError("Key '%s' not in params" % key)
    params["mean"] = [0.485, 0.456, 0.406]
    params["std"] = [0.229, 0.224, 0.225]
    params["architecture"] = params.pop("cir_architecture")
    net = init_network(params)
    net.meta["in_channels"] = 3
    net.meta["out_channels"] = net.meta["outputdim"]
    return CirRetrievalNet(net.features, net.lwhiten, net.pool, net.whiten, net.meta)
