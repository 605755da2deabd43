"""The CPU oracle (oracle/gandtr_oracle.py) against the golden vectors produced by the imported reference
(tests/golden/make_golden.py).  Same torch ops on the same CPU => expected bit-identical; a 1e-6 absolute slack
covers differently vectorised oneDNN paths on another host."""
import os

import numpy as np
import pytest
import torch

from gandtr_amd.tools import synth
from oracle import gandtr_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: np.load(os.path.join(G, name + ".npz"))
TOL = 2e-6


def close(a, b, tol=TOL):
    a, b = torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).abs().max())
    assert err <= tol * max(1.0, float(b.abs().max())), err


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_generator_tiny_every_tap(norm):
    g = load("gen_tiny_" + norm)
    sd = synth.generator_state(0, norm, ngf=8, n_blocks=2)
    x = synth.synth_input(1, (2, 3, 32, 32), 1.0)
    out, feats = O.resnet_generator(x, sd, norm, 2, taps=tuple(range(21)))
    close(out, g["out"])
    for i in range(21):
        close(feats[i], g["tap%d" % i])


@pytest.mark.parametrize("tag,norm,kw", [("in002", "instance", dict(gain=0.02)), ("in02", "instance", dict(gain=0.2)),
                                         ("bn", "batch", {})])
def test_generator_full_size(tag, norm, kw):
    g = load("gen_full_" + tag)
    sd = synth.generator_state(0, norm, **kw)
    x = synth.synth_input(2, (2, 3, 256, 256), 1.0)
    taps = (3, 9, 10, 14, 18, 21, 24, 26, 27)
    _, feats = O.resnet_generator(x, sd, norm, 9, taps=taps)
    for i in taps:
        f = feats[i]
        sub = f[:, ::16, ::16, ::16] if f.shape[1] >= 16 else f[:, :, ::16, ::16]
        close(sub, g["tap%d_sub" % i], 1e-5)
        close(f.mean(dim=(2, 3)), g["tap%d_mean" % i], 1e-5)
        close(f.std(dim=(2, 3)), g["tap%d_std" % i], 1e-5)


def test_gem_l2n():
    g = load("gem_l2n")
    x = torch.from_numpy(g["x"])
    for p in (3.0, 2.37):
        gm = O.gem(x, p=p)
        close(gm, g["gem_p%s" % p])
        close(O.l2n(gm), g["l2n_p%s" % p])


def test_wrappers():
    g = load("wrappers")
    img = synth.synth_input(5, (1, 3, 40, 56))
    for tag, key in (("ms", True), ("sms", "sms")):
        assert np.allclose(g["scales_" + tag], O.SCALE_PRESETS[key])
        for i, lvl in enumerate(O.multiscale_pyramid(img, O.SCALE_PRESETS[key])):
            close(lvl, g["pyr_%s_%d" % (tag, i)])
    vecs = torch.from_numpy(g["vecs"])
    lw = synth.whitening_state(2, 64)
    P, m = torch.from_numpy(lw["P"]), torch.from_numpy(lw["m"])
    for msp in (1.0, 3.0, 2.37):
        agg = O.ms_aggregate(list(vecs), msp)
        close(agg, g["agg_msp%s" % msp])
        for dims in (None, 16):
            close(O.whiten(agg, P, m, dims), g["whiten_msp%s_d%s" % (msp, dims)])
    close(O.meanstd_adapt(img, [[0.5] * 3, [0.5] * 3], [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]), g["meanstd_post"])


@pytest.mark.parametrize("arch,p", [("vgg16", 3.0), ("resnet101", 2.37)])
def test_embedders(arch, p):
    g = load("embed_" + arch)
    sd = synth.vgg16_state(0, p=p) if arch == "vgg16" else synth.resnet101_state(0, p=p)
    x = synth.synth_input(7, (2, 3, 96, 128))
    close(O.image_retrieval_forward(x, sd, arch), g["ss"])
    lw = synth.whitening_state(3, g["ss"].shape[0])
    P, m = torch.from_numpy(lw["P"]), torch.from_numpy(lw["m"])
    for tag, key in (("ms", True), ("sms", "sms")):
        got = torch.stack([O.embed_ms_whiten(x[i:i + 1], sd, arch, O.SCALE_PRESETS[key], P, m) for i in range(2)])
        close(got, g["hub_" + tag], 5e-6)


def test_hed():
    g = load("hed")
    y = synth.synth_input(8, (2, 3, 64, 96), 1.0)
    close(O.hed_on_generator_output(y, synth.hed_state(0)), g["out"])


def test_chain_config5():
    g = load("chain_c5")
    x = synth.synth_input(9, (2, 3, 128, 128), 1.0)
    y = O.resnet_generator(x, synth.generator_state(0, "instance", gain=0.02), "instance", 9)
    y = O.meanstd_adapt(y, [[0.5] * 3, [0.5] * 3], [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]])
    close(O.image_retrieval_forward(y, synth.resnet101_state(0, p=3.0), "resnet101"), g["out"], 5e-6)
