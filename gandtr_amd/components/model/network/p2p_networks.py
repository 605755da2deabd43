"""CycleGAN / HED-N-GAN ResNet generator -- host mirror of mdir/components/model/network/p2p_networks.py
(get_norm_layer :23-35, ResnetGenerator :239-337, ResnetBlock :454-506).

The nn.Module tree is identical to the reference's (same nn.Sequential indices, same parameter names, same creation
order), so reference checkpoints load unchanged and seeded initialisation reproduces the reference's weights.  On a
cuda (HIP) device the forward runs on the hand-written kernels through gandtr_amd.engine; on 'cpu' it runs the stock
torch modules (BASELINE config 0: "plumbing, no GPU").
"""
import functools

import torch
import torch.nn as nn

from ._hipbacked import HipBacked


def get_norm_layer(norm_type="instance", track_running_stats=True):
    """'batch' -> BatchNorm2d(affine, running stats); 'instance' -> InstanceNorm2d(affine=False); 'none' -> identity."""
    if not isinstance(norm_type, str):
        return norm_type
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=track_running_stats)
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=False)
    if norm_type == "none":
        return lambda _channels: nn.Identity()
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def _pad_layer(padding_type):
    """-> (list of explicit padding modules, conv padding) for a 3x3 conv"""
    if padding_type == "reflect":
        return [nn.ReflectionPad2d(1)], 0
    if padding_type == "replicate":
        return [nn.ReplicationPad2d(1)], 0
    if padding_type == "zero":
        return [], 1
    raise NotImplementedError('padding [%s] is not implemented' % padding_type)


class ResnetBlock(nn.Module):
    """x + norm(conv3x3(pad(relu(norm(conv3x3(pad(x)))))));  sub-module indices 0 pad, 1 conv, 2 norm, 3 relu, 4 pad,
    5 conv, 6 norm (dropout, when enabled, shifts the second half by one -- as in the reference)."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        layers = []
        pads, p = _pad_layer(padding_type)
        layers += pads + [nn.Conv2d(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim), nn.ReLU(True)]
        if use_dropout:
            layers.append(nn.Dropout(0.5))
        pads, p = _pad_layer(padding_type)
        layers += pads + [nn.Conv2d(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim)]
        self.conv_block = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv_block(x)


class ResnetGenerator(HipBacked, nn.Module):
    """ResNet generator: 7x7 stem, two stride-2 down convs, n_blocks ResnetBlocks, two ConvTranspose up layers, 7x7
    head + tanh.  Only the hub configuration (no_antialias / no_antialias_up, reflect padding, no dropout) is on the
    HIP hot path; other configurations are rejected with NotImplementedError on a cuda device."""

    #: the generators meet north_star's 1e-3 in the compensated mode only
    hip_default_precision = "f16c"

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer="batch", use_dropout=False, n_blocks=9,
                 padding_type="reflect", no_antialias=True, no_antialias_up=True, track_running_stats=True):
        assert n_blocks >= 0
        super().__init__()
        if not (no_antialias and no_antialias_up):
            raise NotImplementedError("anti-aliased down/up-sampling (CUT) is outside the gandtr hot path")
        self.meta = {"in_channels": input_nc, "out_channels": output_nc}
        self._cfg = dict(norm=norm_layer if isinstance(norm_layer, str) else None, n_blocks=n_blocks,
                         padding_type=padding_type, use_dropout=use_dropout)
        norm_layer = get_norm_layer(norm_layer, track_running_stats)
        base = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
        use_bias = base == nn.InstanceNorm2d

        seq = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias), norm_layer(ngf),
               nn.ReLU(True)]
        ch = ngf
        for _ in range(2):
            seq += [nn.Conv2d(ch, ch * 2, kernel_size=3, stride=2, padding=1, bias=use_bias), norm_layer(ch * 2), nn.ReLU(True)]
            ch *= 2
        for _ in range(n_blocks):
            seq.append(ResnetBlock(ch, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout,
                                   use_bias=use_bias))
        for _ in range(2):
            seq += [nn.ConvTranspose2d(ch, ch // 2, kernel_size=3, stride=2, padding=1, output_padding=1, bias=use_bias),
                    norm_layer(ch // 2), nn.ReLU(True)]
            ch //= 2
        seq += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0), nn.Tanh()]
        self.model = nn.Sequential(*seq)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, input, layers=[], encode_only=False):
        if self._hip_device().type == "cuda":
            return self._forward_hip(input, list(layers), encode_only)
        if -1 in layers:
            layers.append(len(self.model))
        if len(layers) > 0:
            feat, feats = input, []
            for layer_id, layer in enumerate(self.model):
                feat = layer(feat)
                if layer_id in layers:
                    feats.append(feat)
                if layer_id == layers[-1] and encode_only:
                    return feats
            return feat, feats
        return self.model(input)

    def _forward_hip(self, x, layers, encode_only):
        from .... import engine
        cfg = self._cfg
        if cfg["norm"] not in ("instance", "batch") or cfg["padding_type"] != "reflect" or cfg["use_dropout"]:
            raise NotImplementedError("HIP generator supports norm instance|batch, reflect padding, no dropout")
        if self.training and cfg["norm"] == "batch":
            raise NotImplementedError("HIP generator is inference-only: call .eval() (BatchNorm uses running statistics)")
        last = len(self.model) - 1
        taps = tuple(sorted({l for l in layers if l != -1 and l <= last}))
        unavailable = [t for t in taps if t in (0, last - 2)]
        if unavailable:
            raise NotImplementedError("feature taps %s (reflection-padded tensors) are not materialised on the HIP path" % unavailable)
        self._hip_check_inference()
        prec = self._hip_precision()
        net = self._hip_net(("gen", taps, prec), lambda sd, dev: engine.build_generator(sd, dev, taps=taps, precision=prec, norm=cfg["norm"]))
        outs = net.forward(x)
        out = outs[net.out_slot]
        if not layers:
            return out
        feats = [outs[net.tap_slots[t]] for t in layers if t in net.tap_slots]
        if encode_only and layers[-1] in net.tap_slots:
            return feats
        return out, feats
