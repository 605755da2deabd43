"""GPU path against the committed golden vectors (outputs of the imported reference, tests/golden/make_golden.py) and the
hub / wrapper plumbing on a cuda device.  The hub generators run their default "f16c" precision: 1e-3 against the reference."""
import os
import pickle

import numpy as np
import pytest
import torch

import hubconf
from gandtr_amd.learning import network as N
from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: np.load(os.path.join(G, name + ".npz"))


def rel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).float(), torch.as_tensor(np.asarray(b)).float()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("norm", ["instance", "batch"])
def test_generator_tiny_taps_vs_reference(cuda_device, norm):
    g = load("gen_tiny_" + norm)
    from gandtr_amd.components.model import network as M
    gen = M.initialize_model({"architecture": "official_resnet_generator", "input_nc": 3, "output_nc": 3, "ngf": 8,
                              "n_blocks": 2, "norm_layer": norm}).eval()
    gen.load_state_dict(synth.generator_state(0, norm, ngf=8, n_blocks=2))
    gen.to(cuda_device)
    taps = [1, 2, 3, 4, 6, 7, 9, 10, 11, 12, 14, 15, 17, 19, 20]
    with torch.no_grad():
        out, feats = gen(synth.synth_input(1, (2, 3, 32, 32), 1.0).to(cuda_device), layers=list(taps))
    assert rel(out.cpu(), g["out"]) < 1e-3
    for t, f in zip(taps, feats):
        assert rel(f.cpu(), g["tap%d" % t]) < 1e-3, t


@pytest.mark.parametrize("name", ["cyclegan", "hedngan"])
def test_hub_generator_on_gpu(cuda_device, name):
    """hub entrypoint on the GPU: same seeded weights as the reference, output vs the reference's CPU output"""
    g = load("hub_" + name)
    net = getattr(hubconf, name)(pretrained=False, device=cuda_device)
    x = synth.synth_input(3, (4, 3, 256, 256), 1.0)
    with torch.no_grad():
        y = net(x)
    assert y.is_cuda and y.shape == (4, 3, 256, 256)
    y = y.cpu()
    # the reference's own output on the same seed-0 weights (tests/golden/make_golden.py).  These fixtures are SATURATED: cyclegan's gain-0.2
    # init drives |pre-tanh| to ~35, hedngan's kaiming / BatchNorm init to several hundred, so north_star's 1e-3 of the pre-tanh
    # range is 3.5e-2 / ~0.3 in front of the tanh and outputs near a zero crossing move by that much while the image as a whole agrees to
    # < 5e-4 in the mean.  All three statistics are gated, each with its measured value next to it (tools/parity_report.py, round 3):
    #   cyclegan  mean 2.97e-4   p99.9 7.5e-3   max 9.6e-3        hedngan  mean 4.93e-4   p99.9 0.100   max 0.259
    # (single-pass fp16 mode: mean 1.8e-3; the unsaturated absolute gate, max|d image| <= 1e-3, is test_hip_models.py::test_generator_image_absolute_gate)
    diff = (y[:, :, ::8, ::8] - torch.from_numpy(g["out_sub"])).abs().flatten()
    mean, p999, mx = float(diff.mean()), float(torch.quantile(diff, 0.999)), float(diff.max())
    print("hub %s vs the reference's output: mean %.2e, p99.9 %.2e, max %.2e" % (name, mean, p999, mx))
    gate = {"cyclegan": (5e-4, 1.5e-2, 3.5e-2), "hedngan": (1e-3, 0.2, 0.5)}[name]
    assert mean < gate[0] and p999 < gate[1] and mx < gate[2], (mean, p999, mx)


@pytest.mark.parametrize("arch,p", [("vgg16", 3.0), ("resnet101", 2.37)])
def test_embedder_hub_paths_on_gpu(cuda_device, arch, p, tmp_path):
    g = load("embed_" + arch)
    state = synth.vgg16_state(0, p=p) if arch == "vgg16" else synth.resnet101_state(0, p=p)
    params = {"type": "SingleNetwork",
              "model": {"architecture": "cirnet", "cir_architecture": arch, "local_whitening": False, "pooling": "gem",
                        "pretrained": False, "regional": False, "whitening": False},
              "initialize": False,
              "runtime": {"data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]},
                          "wrappers": "cirfaketuplebatch"}}
    net = N.initialize_network(params, cuda_device).eval()
    net.model.load_state_dict(state)
    x = synth.synth_input(7, (2, 3, 96, 128))
    with torch.no_grad():
        d = net(x)
    assert d.is_cuda and d.shape == g["ss"].shape                     # D x N
    ref = torch.from_numpy(g["ss"])
    assert float((d.cpu() - ref).abs().max()) < 1e-3
    assert float(torch.nn.functional.cosine_similarity(d.cpu().t(), ref.t(), dim=1).min()) > 0.9999
    # pretrained-style path: whiten + multiscale wrappers, pyramid fused into the HIP pack kernel
    ck = tmp_path / "net.pth"
    cpu_sd = {"type": "SingleNetwork", "frozen": False, "network_params": net.network_params._asdict(),
              "model_state": {k: v.cpu() for k, v in net.model.state_dict().items()}}
    torch.save(cpu_sd, ck)
    lwp = tmp_path / "lw.pkl"
    with open(lwp, "wb") as f:
        pickle.dump(synth.whitening_state(3, g["ss"].shape[0]), f)
    from gandtr_amd.learning.checkpoints import Checkpoints
    for tag, scales in (("ms", True), ("sms", "sms")):
        runtime = {"wrappers": {"train": None, "eval": {"0_cirwhiten": {"whitening": str(lwp), "dimensions": None},
                                                        "1_cirmultiscale": {"scales": scales}}}}
        hub = N.initialize_network(None, cuda_device, Checkpoints.load_network(str(ck)), runtime).eval()
        refh = torch.from_numpy(g["hub_" + tag])                      # 2 x D
        with torch.no_grad():
            one = hub(x[0:1].clone())
            both = hub(x.clone())
        assert one.shape == (refh.shape[1],) and both.shape == (refh.shape[1], 2)
        assert float((one.cpu() - refh[0]).abs().max()) < 1e-3
        assert float((both.cpu().t() - refh).abs().max()) < 1e-3
        assert float(torch.nn.functional.cosine_similarity(both.cpu().t(), refh, dim=1).min()) > 0.9999


def test_hed_with_wrappers_on_gpu(cuda_device, monkeypatch):
    g = load("hed")
    params = {"type": "SingleNetwork", "model": {"architecture": "hed_interpolation"}, "initialize": False,
              "runtime": {"wrappers": "rgb2bgr_pre, meanstd_pre:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:[[0.40787054,0.45752458,0.48109378],[1,1,1]]"}}
    net = N.initialize_network(params, cuda_device).eval()
    net.model.load_state_dict(synth.hed_state(0))
    # on a HIP device both wrappers are folded into the HED net's input-pack kernel (no torch op on the data path): their own
    # preprocess must not run
    for w in net.wrappers["eval"].wrappers:
        monkeypatch.setattr(w, "preprocess", lambda *a, **k: (_ for _ in ()).throw(AssertionError("wrapper ran as a torch op")))
    with torch.no_grad():
        out = net(synth.synth_input(8, (2, 3, 64, 96), 1.0).to(cuda_device))   # the generator output lives on the device
    assert float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) < 1e-3
    keys = [k for k in net.model._hip_cache if isinstance(k, tuple) and k[0] == "hed"]
    assert keys and keys[0][3] is not None and keys[0][3][0] == (2, 1, 0)
    assert not hasattr(net.model, "input_transform")                            # a call argument, never module state


def test_chain_config5_on_gpu(cuda_device):
    g = load("chain_c5")
    gen = {"type": "SingleNetwork",
           "model": {"architecture": "official_resnet_generator", "input_nc": 3, "output_nc": 3, "n_blocks": 9,
                     "norm_layer": "instance", "no_antialias": True, "no_antialias_up": True},
           "initialize": False,
           "runtime": {"wrappers": "meanstd_post:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:[[0.485,0.456,0.406],[0.229,0.224,0.225]]",
                       "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    emb = {"type": "SingleNetwork",
           "model": {"architecture": "cirnet", "cir_architecture": "resnet101", "local_whitening": False, "pooling": "gem",
                     "pretrained": False, "regional": False, "whitening": False},
           "initialize": False,
           "runtime": {"wrappers": "cirfaketuplebatch",
                       "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    chain = N.initialize_network({"type": "CirSequentialNetwork", "sequence": "augment,embed", "augment": gen, "embed": emb},
                                 cuda_device).eval()
    chain.networks["augment"].model.load_state_dict(synth.generator_state(0, "instance", gain=0.02))
    chain.networks["embed"].model.load_state_dict(synth.resnet101_state(0, p=3.0))
    with torch.no_grad():
        d = chain(synth.synth_input(9, (2, 3, 128, 128), 1.0))
    ref = torch.from_numpy(g["out"])
    assert d.shape == ref.shape
    assert float(torch.nn.functional.cosine_similarity(d.cpu().t(), ref.t(), dim=1).min()) > 0.9999
    assert float((d.cpu() - ref).abs().max()) < 1e-3


def test_hub_generator_precision_switch(cuda_device, monkeypatch):
    """GANDTR_HIP_PRECISION / Module.hip_precision select the conv arithmetic on the hub path: the default (f16c) and f16x3 are
    far closer to the reference than the opt-in single-pass fp16"""
    g = load("hub_cyclegan")
    ref = torch.from_numpy(g["out_sub"])
    x = synth.synth_input(3, (4, 3, 256, 256), 1.0)
    net = hubconf.cyclegan(pretrained=False, device=cuda_device)
    assert net.model._hip_precision() == "f16c"
    with torch.no_grad():
        default = net(x).cpu()[:, :, ::8, ::8]
        net.model.hip_precision = "f16"
        fast = net(x).cpu()[:, :, ::8, ::8]
        net.model.hip_precision = "f16x3"
        exact = net(x).cpu()[:, :, ::8, ::8]
    e_default, e_fast, e_exact = (float((t - ref).abs().mean()) for t in (default, fast, exact))
    # measured (round 3): f16c 2.97e-4, f16 1.8e-3, f16x3 1e-5 -- mean |d image| on the saturated seed-0 fixture (see test_hub_generator_on_gpu)
    assert e_exact < 2e-5 and e_default < 3.5e-4 and e_default < e_fast / 5, (e_default, e_fast, e_exact)
