"""Deterministic weight / input synthesiser.

There is no network on the build or GPU boxes, so benchmarks and parity tests run on random-initialised
weights of the reference architectures.  Everything here is generated with a counter-based NumPy Philox
stream keyed by (seed, crc32(parameter name)), so the build container and the GPU box regenerate
bit-identical tensors without shipping them (never rely on reference-side RNG).

State-dict key names follow the reference modules:
  generator  -- mdir/components/model/network/p2p_networks.py:269-313 (``model.<i>...``)
  embedders  -- torchvision vgg16/resnet101 as sliced by external/cirtorch/networks/imageretrievalnet.py:185-190
                (``features.<i>...``) plus ``pool.p`` (layers/pooling.py:40)
  HED        -- mdir/components/model/network/hed.py:30-45
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std=1.0, mean=0.0):
    a = _rng(seed, name).standard_normal(size=shape, dtype=np.float32)
    return torch.from_numpy(a * np.float32(std) + np.float32(mean))


def _uniform(seed, name, shape, lo, hi):
    a = _rng(seed, name).random(size=shape, dtype=np.float32)
    return torch.from_numpy(a * np.float32(hi - lo) + np.float32(lo))


def synth_input(seed, shape, clamp=None, name="input"):
    """Seeded N(0,1) input; generator inputs are clamped to [-1, 1] (mean/std 0.5 normalised images)."""
    x = _normal(seed, name, shape)
    return x.clamp_(-clamp, clamp) if clamp else x


def _conv(sd, seed, name, cout, cin, k, bias, gain=None, transposed=False):
    fan_in = cin * k * k
    std = gain if gain is not None else math.sqrt(2.0 / fan_in)
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    sd[name + ".weight"] = _normal(seed, name + ".weight", shape, std)
    if bias:
        sd[name + ".bias"] = _normal(seed, name + ".bias", (cout,), 0.1)


def _bn(sd, seed, name, c, gamma_scale=1.0):
    sd[name + ".weight"] = _uniform(seed, name + ".weight", (c,), 0.8, 1.2) * gamma_scale
    sd[name + ".bias"] = _normal(seed, name + ".bias", (c,), 0.1)
    sd[name + ".running_mean"] = _normal(seed, name + ".running_mean", (c,), 0.1)
    sd[name + ".running_var"] = _uniform(seed, name + ".running_var", (c,), 0.8, 1.2)
    sd[name + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.long)


def generator_state(seed=0, norm="instance", ngf=64, n_blocks=9, in_nc=3, out_nc=3, gain=0.02):
    """ResnetGenerator state dict (p2p_networks.py:269-313).  ``norm='instance'``: N(0, gain) conv weights with
    biases (use_bias, :264-267); ``norm='batch'``: kaiming conv weights, no conv bias, non-trivial BN running
    statistics (last BN of every ResnetBlock scaled by 0.5 so nine residual blocks stay O(1))."""
    sd = {}
    inorm = norm == "instance"
    g = gain if inorm else None

    def norm_at(name, c, scale=1.0):
        if not inorm:
            _bn(sd, seed, name, c, scale)

    _conv(sd, seed, "model.1", ngf, in_nc, 7, inorm, g)
    norm_at("model.2", ngf)
    i, c = 4, ngf
    for _ in range(2):
        _conv(sd, seed, "model.%d" % i, 2 * c, c, 3, inorm, g)
        norm_at("model.%d" % (i + 1), 2 * c)
        i, c = i + 3, 2 * c
    for _ in range(n_blocks):
        p = "model.%d.conv_block." % i
        _conv(sd, seed, p + "1", c, c, 3, inorm, g)
        norm_at(p + "2", c)
        _conv(sd, seed, p + "5", c, c, 3, inorm, g)
        norm_at(p + "6", c, 0.5)
        i += 1
    for _ in range(2):
        _conv(sd, seed, "model.%d" % i, c // 2, c, 3, inorm, g, transposed=True)
        norm_at("model.%d" % (i + 1), c // 2)
        i, c = i + 3, c // 2
    _conv(sd, seed, "model.%d" % (i + 1), out_nc, c, 7, True, g)
    return sd


VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]


def vgg16_state(seed=0, p=3.0, width_div=1):
    """GeM-VGG16 embedder state dict: ``features.{0,2,5,...,28}.{weight,bias}`` + ``pool.p``."""
    sd, i, cin = {}, 0, 3
    for v in VGG16_CFG:
        if v == "M":
            i += 1
            continue
        _conv(sd, seed, "features.%d" % i, v // width_div, cin, 3, True)
        cin, i = v // width_div, i + 2
    sd["pool.p"] = torch.ones(1) * p
    return sd


def resnet101_state(seed=0, p=3.0, blocks=(3, 4, 23, 3), width_div=1):
    """GeM-ResNet-101 embedder state dict: ``features.0`` conv1, ``features.1`` bn1, ``features.{4..7}.<b>.*``
    Bottlenecks (bn3 gamma x0.25 so the residual stream stays O(1) without calibrated statistics) + ``pool.p``."""
    sd = {}
    base = 64 // width_div
    _conv(sd, seed, "features.0", base, 3, 7, False)
    _bn(sd, seed, "features.1", base)
    inpl = base
    for li, nb in enumerate(blocks):
        planes = base * (2 ** li)
        for b in range(nb):
            q = "features.%d.%d." % (4 + li, b)
            _conv(sd, seed, q + "conv1", planes, inpl, 1, False)
            _bn(sd, seed, q + "bn1", planes)
            _conv(sd, seed, q + "conv2", planes, planes, 3, False)
            _bn(sd, seed, q + "bn2", planes)
            _conv(sd, seed, q + "conv3", planes * 4, planes, 1, False)
            _bn(sd, seed, q + "bn3", planes * 4, 0.25)
            if b == 0:
                _conv(sd, seed, q + "downsample.0", planes * 4, inpl, 1, False)
                _bn(sd, seed, q + "downsample.1", planes * 4)
            inpl = planes * 4
    sd["pool.p"] = torch.ones(1) * p
    return sd


HED_BLOCKS = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))


def hed_state(seed=0, width_div=1):
    """HedInterpolation state dict (hed.py:30-45)."""
    sd, cin = {}, 3
    for bi, chans in enumerate(HED_BLOCKS):
        off = 0 if bi == 0 else 1
        for ci, c in enumerate(chans):
            _conv(sd, seed, "vgg%d.%d" % (bi + 1, off + 2 * ci), c // width_div, cin, 3, True)
            cin = c // width_div
        _conv(sd, seed, "score%d" % (bi + 1), 1, cin, 1, True, gain=1.0 / math.sqrt(cin))
    _conv(sd, seed, "fusion.0", 1, 5, 1, True, gain=0.5)
    return sd


def whitening_state(seed, dim):
    """Synthetic learned-whitening ``{'P': DxD, 'm': Dx1}`` (format: wrapper.py:315-317, stages/whiten.py:75):
    a well-conditioned random matrix (random diagonal + small dense part; elementwise-deterministic) and a small mean."""
    a = _rng(seed, "lw.P").standard_normal(size=(dim, dim), dtype=np.float32)
    scale = _rng(seed, "lw.s").random(size=(dim,), dtype=np.float32) * np.float32(1.5) + np.float32(0.5)
    P = (a * np.float32(0.3 / math.sqrt(dim)) + np.diag(scale)).astype(np.float32)
    m = (_rng(seed, "lw.m").standard_normal(size=(dim, 1)) * 0.01).astype(np.float32)
    return {"P": P, "m": m}
