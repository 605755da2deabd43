"""Host-side mirror of the reference interface on CPU: hub entrypoints (reference test: test/integration/hub/
test_hub_init.py:17-20), registries, wrappers, containers -- checked against golden vectors from the imported reference."""
import os
import pickle

import numpy as np
import pytest
import torch

import hubconf
from gandtr_amd.components.data import wrapper as W
from gandtr_amd.components.model import network as M
from gandtr_amd.learning import network as N
from gandtr_amd.tools import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: np.load(os.path.join(G, name + ".npz"))

ENTRYPOINTS = ["gem_vgg16_cyclegan", "gem_resnet101_cyclegan", "gem_vgg16_hedngan", "gem_resnet101_hedngan", "cyclegan",
               "hedngan"]


def close(a, b, tol=2e-6):
    a, b = torch.as_tensor(np.asarray(a)), torch.as_tensor(np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_can_initialize_base_models():
    """== reference test_can_initialize_base_models; plus the attribute surface listed in SURVEY.md section 8b"""
    for name in ENTRYPOINTS:
        net = getattr(hubconf, name)(pretrained=False, device="cpu")
        assert net is not None and net.stage == "eval" and not net.model.training
        for attr in ("model", "wrappers", "network_params", "meta", "stage", "device", "frozen", "transform", "forward",
                     "forward_batch", "eval", "train", "freeze", "parameters", "state_dict", "overlay_params", "overlay_model"):
            assert hasattr(net, attr), (name, attr)
        assert set(net.wrappers) == {"train", "eval"}
        assert set(net.state_dict()["net"]) == {"type", "frozen", "network_params", "model_state"}
    assert hubconf.dependencies


def test_pretrained_needs_network():
    """pretrained=True downloads from ptak.felk.cvut.cz (mdir/hub/model.py:5); offline it must raise, not fall back"""
    with pytest.raises(Exception):
        hubconf.cyclegan(pretrained=True, device="cpu")


@pytest.mark.parametrize("name", ["cyclegan", "hedngan"])
def test_hub_generator_reproduces_reference_init_and_output(name):
    """seed-0 normal_p2p / kaiming_p2p initialisation (network.py:159-162) gives the reference's weights bit for bit, and
    the CPU forward (BASELINE config 0: 4x3x256x256) gives the reference's output"""
    g = load("hub_" + name)
    net = getattr(hubconf, name)(pretrained=False, device="cpu")
    sd = net.model.state_dict()
    assert sorted(sd.keys()) == list(g["keys"])
    for k, s, a in zip(g["keys"], g["wsum"], g["wabs"]):
        assert float(sd[k].double().sum()) == pytest.approx(s, rel=1e-12, abs=1e-12), k
        assert float(sd[k].double().abs().sum()) == pytest.approx(a, rel=1e-12, abs=1e-12), k
    x = synth.synth_input(3, (4, 3, 256, 256), 1.0)
    with torch.no_grad():
        y = net(x)
    close(y[:, :, ::8, ::8], g["out_sub"], 1e-5)
    close(y.mean(dim=(2, 3)), g["out_mean"], 1e-5)


def test_transform_repr_matches_readme():
    """README.md:132-138"""
    net = hubconf.gem_vgg16_hedngan(pretrained=False, device="cpu")
    assert repr(net.transform) == ("Compose(\n    Pil2Numpy()\n    ApplyClahe(clip_limit=1.0, grid_size=8, colorspace=lab)\n"
                                   "    ToTensor()\n    Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], "
                                   "strict_shape=True)\n)")
    gen = hubconf.cyclegan(pretrained=False, device="cpu")
    img = (np.random.RandomState(0).rand(8, 9, 3) * 255).astype(np.uint8)
    t = gen.transform(img)
    assert t.shape == (3, 8, 9) and float(t.min()) >= -1.0 and float(t.max()) <= 1.0


def test_registries_and_errors():
    assert {"official_resnet_generator", "cirnet", "hed_interpolation"} <= set(M.MODEL_LABELS)
    assert {"cirmultiscale", "cirwhiten", "cirfaketuplebatch", "fakebatch", "meanstd_post", "meanstd_pre", "rgb2bgr_pre",
            "clahepost"} <= set(W.WRAPPERS_LABELS)
    assert {"SingleNetwork", "CirSequentialNetwork"} <= set(N.NETWORKS)
    with pytest.raises(KeyError):
        M.initialize_model({"architecture": "no_such_model"})
    with pytest.raises(KeyError):
        W.initialize_wrappers("no_such_wrapper", "cpu")
    with pytest.raises(ValueError):
        M.initialize_model({"architecture": "cirnet", "cir_architecture": "vgg16"})          # missing keys (cirnet.py:49-51)
    with pytest.raises(ValueError):
        M.initialize_model({"architecture": "cirnet", "cir_architecture": "alexnet9", "local_whitening": False,
                            "pooling": "gem", "regional": False, "whitening": False, "pretrained": False})
    with pytest.raises(NotImplementedError):
        M.initialize_model({"architecture": "official_resnet_generator", "input_nc": 3, "output_nc": 3, "norm_layer": "group"})
    with pytest.raises(AssertionError):      # unknown runtime key (network.py:128-131)
        N.initialize_network({"type": "SingleNetwork", "model": {"architecture": "identity"}, "initialize": False,
                              "runtime": {"wrappers": "", "bogus": 1}}, "cpu")


def test_wrapper_order_and_dsl():
    w = W.initialize_wrappers({"1_cirmultiscale": {"scales": True}, "0_cirwhiten":
                               {"whitening": synth.whitening_state(0, 8), "dimensions": None}}, "cpu")
    assert [type(x).__name__ for x in w.wrappers] == ["CirtorchWhiten", "CirMultiscaleAggregation"]
    w = W.initialize_wrappers("rgb2bgr_pre, meanstd_pre:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:[[0.4,0.45,0.48],[1,1,1]]", "cpu")
    assert [type(x).__name__ for x in w.wrappers] == ["RgbToBgrPre", "MeanStdPre"]
    assert W.CirMultiscaleAggregation("sms", "cpu").scales == [1, 1. / np.sqrt(2), np.sqrt(2)]
    assert W.CirMultiscaleAggregation(True, "cpu").scales == [1, 1. / np.sqrt(2), 1. / 2]


def test_wrappers_against_reference_vectors():
    g = load("wrappers")
    img = synth.synth_input(5, (1, 3, 40, 56))
    for tag, sc in (("ms", "True"), ("sms", "sms")):
        levels, waslist = W.CirMultiscaleAggregation(sc, "cpu").preprocess(img, None)
        assert not waslist
        for i, lvl in enumerate(levels):
            close(lvl, g["pyr_%s_%d" % (tag, i)])
    vecs = torch.from_numpy(g["vecs"])
    lw = synth.whitening_state(2, 64)
    for msp in (1.0, 3.0, 2.37):
        agg = W.CirMultiscaleAggregation.aggregate_tensor([v.clone() for v in vecs], 3, 64, msp)
        close(agg, g["agg_msp%s" % msp])
        for dims in (None, 16):
            close(W.CirtorchWhiten(lw, dims, "cpu").postprocess(agg.clone(), None, None), g["whiten_msp%s_d%s" % (msp, dims)])
    ms = W.MeanStdPost("[[0.5,0.5,0.5],[0.5,0.5,0.5]]", "[[0.485,0.456,0.406],[0.229,0.224,0.225]]", device="cpu")
    close(ms.postprocess(img.clone(), None, None), g["meanstd_post"])
    sc, sh = ms.affine()
    close(img * torch.tensor(sc)[None, :, None, None] + torch.tensor(sh)[None, :, None, None], g["meanstd_post"], 1e-5)


def _embedder(arch, state, wrappers):
    params = {"type": "SingleNetwork",
              "model": {"architecture": "cirnet", "cir_architecture": arch, "local_whitening": False, "pooling": "gem",
                        "pretrained": False, "regional": False, "whitening": False},
              "initialize": False,
              "runtime": {"data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]},
                          "wrappers": wrappers}}
    net = N.initialize_network(params, "cpu").eval()
    net.model.load_state_dict(state)
    return net


@pytest.mark.parametrize("arch,p", [("vgg16", 3.0), ("resnet101", 2.37)])
def test_embedder_paths(arch, p, tmp_path):
    g = load("embed_" + arch)
    state = synth.vgg16_state(0, p=p) if arch == "vgg16" else synth.resnet101_state(0, p=p)
    x = synth.synth_input(7, (2, 3, 96, 128))
    net = _embedder(arch, state, "cirfaketuplebatch")
    with torch.no_grad():
        close(net(x), g["ss"])                                              # D x N
    # pretrained-style path: checkpoint file + lw.pkl -> whiten + multiscale wrappers (hub/model.py:30-34)
    ck = tmp_path / "net.pth"
    torch.save(net.state_dict()["net"], ck)
    lwp = tmp_path / "lw.pkl"
    with open(lwp, "wb") as f:
        pickle.dump(synth.whitening_state(3, g["ss"].shape[0]), f)
    from gandtr_amd.learning.checkpoints import Checkpoints
    for tag, scales in (("ms", True), ("sms", "sms")):
        runtime = {"wrappers": {"train": None, "eval": {"0_cirwhiten": {"whitening": str(lwp), "dimensions": None},
                                                        "1_cirmultiscale": {"scales": scales}}}}
        hub = N.initialize_network(None, "cpu", Checkpoints.load_network(str(ck)), runtime).eval()
        with torch.no_grad():
            single = torch.stack([hub(x[i:i + 1].clone()) for i in range(2)])
            batched = hub(x.clone())                                        # extension: batch > 1 == stack of singles (D4)
        close(single, g["hub_" + tag], 5e-6)
        assert single[0].shape == (g["ss"].shape[0],)
        close(batched.t(), g["hub_" + tag], 5e-6)


def test_hed_with_wrappers():
    g = load("hed")
    params = {"type": "SingleNetwork", "model": {"architecture": "hed_interpolation"}, "initialize": False,
              "runtime": {"wrappers": "rgb2bgr_pre, meanstd_pre:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:[[0.40787054,0.45752458,0.48109378],[1,1,1]]"}}
    net = N.initialize_network(params, "cpu").eval()
    net.model.load_state_dict(synth.hed_state(0))
    with torch.no_grad():
        close(net(synth.synth_input(8, (2, 3, 64, 96), 1.0)), g["out"])


def test_input_wrappers_fold_into_one_transform():
    """Compose hands the trailing per-channel input wrappers to a HIP model as y[c] = x[perm[c]] * scale[c] + shift[c]
    (RgbToBgrPre then MeanStdPre: mdir/components/data/wrapper.py:351-364, :182-194); on the CPU nothing is folded."""
    from gandtr_amd.components.data.wrapper import initialize_wrappers

    class Fake:
        accepts_input_transform = True
        meta = {"in_channels": 3, "out_channels": 1}

        def __init__(self, kind):
            self.kind = kind

        def _hip_device(self):
            return torch.device(self.kind)

    chain = initialize_wrappers("rgb2bgr_pre, meanstd_pre:[[0.5,0.5,0.5],[0.5,0.25,0.5]]:[[0.4,0.45,0.48],[1,2,1]]", "cpu")
    active, folded = chain._fold_input_wrappers(Fake("cuda"))
    assert active == [] and folded[0] == (2, 1, 0)
    x = torch.rand(2, 3, 4, 5)
    ref = x
    for w in chain.wrappers:
        ref, _ = w.preprocess(ref, None)
    perm, scale, shift = folded
    got = torch.stack([x[:, perm[c]] * scale[c] + shift[c] for c in range(3)], 1)
    assert torch.allclose(got, ref, atol=1e-6)
    active, folded = chain._fold_input_wrappers(Fake("cpu"))
    assert len(active) == 2 and folded is None
    # a wrapper with a postprocess of its own in last position blocks the fold
    mixed = initialize_wrappers("rgb2bgr_pre, fakebatch", "cpu")
    assert mixed._fold_input_wrappers(Fake("cuda"))[1] is None
    # mean / std given as one broadcast value per side (mean2tensor accepts it) is not a per-channel list of the model's channel
    # count: not folded, the wrapper runs on the host as in the reference
    scalarish = initialize_wrappers("meanstd_pre:[[0.5],[0.5]]:[[0.4],[1]]", "cpu")
    assert scalarish._fold_input_wrappers(Fake("cuda"))[1] is None
    # only SingleNetwork.forward asks for the fold; a container whose hoisted wrappers act on the chain input does not
    seen = {}

    def inference(t, **kw):
        seen.update(kw)
        return t
    chain(torch.rand(1, 3, 4, 4), inference, outputmodel=Fake("cuda"))
    assert "input_transform" not in seen
    chain(torch.rand(1, 3, 4, 4), inference, outputmodel=Fake("cuda"), fold_input=True)
    assert seen["input_transform"][0] == (2, 1, 0)


def test_cir_sequential_chain_config5():
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
                                 "cpu").eval()
    chain.networks["augment"].model.load_state_dict(synth.generator_state(0, "instance", gain=0.02))
    chain.networks["embed"].model.load_state_dict(synth.resnet101_state(0, p=3.0))
    with torch.no_grad():
        close(chain(synth.synth_input(9, (2, 3, 128, 128), 1.0)), g["out"], 5e-6)


def test_generator_taps_cpu():
    """feature taps (p2p_networks.py:316-334) incl. the in-place ReLU aliasing, against the reference vectors"""
    g = load("gen_tiny_instance")
    gen = M.initialize_model({"architecture": "official_resnet_generator", "input_nc": 3, "output_nc": 3, "ngf": 8,
                              "n_blocks": 2, "norm_layer": "instance"}).eval()
    gen.load_state_dict(synth.generator_state(0, "instance", ngf=8, n_blocks=2))
    with torch.no_grad():
        out, feats = gen(synth.synth_input(1, (2, 3, 32, 32), 1.0), layers=[1, 2, 3, 10, 19])
    close(out, g["out"])
    for f, i in zip(feats, [1, 2, 3, 10, 19]):
        close(f, g["tap%d" % i])


def test_mdir_alias_is_a_drop_in_import():
    import mdir                                              # noqa: F401
    from mdir.hub.model import cyclegan as c2
    from mdir.components.model.network import MODEL_LABELS
    from mdir.components.data.wrapper import WRAPPERS_LABELS
    from mdir.learning.network import NETWORKS, initialize_network
    from mdir.learning import load_network
    import mdir.stages.infer as stage
    assert c2 is hubconf.cyclegan and "cirnet" in MODEL_LABELS and "cirwhiten" in WRAPPERS_LABELS
    assert "SingleNetwork" in NETWORKS and callable(initialize_network) and callable(load_network) and callable(stage.infer)


def test_infer_stage_contract(tmp_path):
    """stage ABI: fn(params, data) -> (metadata, *outputs); embedding and rgb sinks, network given by path or by params"""
    from gandtr_amd.stages import FUNCTIONS
    infer = FUNCTIONS["mdir.stages.infer.infer"]
    emb = {"type": "SingleNetwork",
           "model": {"architecture": "cirnet", "cir_architecture": "vgg16", "local_whitening": False, "pooling": "gem",
                     "pretrained": False, "regional": False, "whitening": False},
           "initialize": False, "path": None,
           "runtime": {"wrappers": "cirfaketuplebatch",
                       "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    imgs = [synth.synth_input(20 + i, (3, 64, 80)) for i in range(3)]
    meta, vecs = infer({"network": emb, "output": {"inference": {"name": "embedding"}}}, (imgs,))
    assert vecs.shape == (3, 512) and np.allclose(np.linalg.norm(vecs, axis=1), 1.0, atol=1e-4) and meta["stats"]["items"] == 3
    gen = hubconf.cyclegan(pretrained=False, device="cpu")
    ck = tmp_path / "gen.pth"
    sd = gen.state_dict()["net"]
    sd["network_params"]["runtime"]["data"] = {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}
    torch.save(sd, ck)
    meta, pics = infer({"network": {"path": str(ck), "runtime": {"wrappers": ""}}, "output": {"inference": {"name": "rgb"}}},
                       ([synth.synth_input(30, (3, 32, 32), 1.0)],))
    assert len(pics) == 1 and pics[0].shape == (32, 32, 3) and 0.0 <= pics[0].min() and pics[0].max() <= 1.0
    assert infer({"network": emb, "output": {"inference": {"name": "embedding"}}}, ([],)) == ({"status": "skipped"},)
    # the HIP device's grouped path (equal sizes as one batch, outputs in input order), exercised here on the CPU against the item-by-item loop
    from gandtr_amd.stages.infer import _infer_grouped
    import gandtr_amd.learning as L
    import copy
    net = L.load_network({"path": str(ck), "runtime": {"wrappers": ""}}, "cpu").eval()
    mixed = [synth.synth_input(60 + i, (3, 32, 32) if i % 2 else (3, 32, 48), 1.0) for i in range(5)]
    _, loop = infer({"network": {"path": str(ck), "runtime": {"wrappers": ""}}, "output": {"inference": {"name": "rgb"}}}, (mixed,))
    _, grouped = _infer_grouped(net, mixed, "rgb", [[0.5] * 3, [0.5] * 3], torch.device("cpu"), 4, 0.0)
    assert len(grouped) == 5 and all(g.shape == l.shape and np.abs(g - l).max() < 1e-5 for g, l in zip(grouped, loop))
    enet = L.load_network(copy.deepcopy(emb), "cpu").eval()
    _, gv = _infer_grouped(enet, imgs, "embedding", None, torch.device("cpu"), 4, 0.0)
    assert gv.shape == (3, 512) and np.allclose(np.linalg.norm(gv, axis=1), 1.0, atol=1e-4)


def test_validate_stage_contract():
    """validate(params, data) -> (metadata,) as the reference's stage (mdir/stages/validate.py:15-39: a 1-tuple; dataset / criterion
    parameters it cannot honour are rejected, not ignored); rank_images -> (metadata, ranks, scores): descriptor extraction + the
    reference's two scoring lines (cirscore.py:51-73, imageretrievalnet.py:312-339); without queries the database queries itself"""
    from gandtr_amd.stages import FUNCTIONS
    from gandtr_amd.stages.validate import extract_vectors
    import mdir.stages.validate as aliased
    validate = FUNCTIONS["mdir.stages.validate.validate"]
    rank_images = FUNCTIONS["gandtr_amd.stages.validate.rank_images"]
    assert aliased.validate is validate
    emb = {"type": "SingleNetwork",
           "model": {"architecture": "cirnet", "cir_architecture": "vgg16", "local_whitening": False, "pooling": "gem",
                     "pretrained": False, "regional": False, "whitening": False},
           "initialize": False, "path": None,
           "runtime": {"wrappers": "cirfaketuplebatch",
                       "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    db = [synth.synth_input(40 + i, (3, 64, 80 - 8 * (i % 2))) for i in range(5)]      # mixed sizes: one forward per image
    qs = [db[3] + 0.01 * synth.synth_input(50, db[3].shape), db[0]]
    params = {"network": emb, "validation": {}, "data": {}}
    import copy
    out = validate(copy.deepcopy(params), (db, qs))
    assert isinstance(out, tuple) and len(out) == 1 and set(out[0]) == {"eval", "retrieval"}          # the reference's arity
    meta, ranks, scores = rank_images({"network": copy.deepcopy(emb)}, (db, qs))
    assert np.array_equal(out[0]["retrieval"]["ranks"], ranks) and np.array_equal(out[0]["retrieval"]["scores"], scores)
    assert ranks.shape == (5, 2) and scores.shape == (5, 2) and meta["eval"] == {**meta["eval"], "database": 5, "queries": 2, "dim": 512}
    assert ranks[0, 0] == 3 and ranks[0, 1] == 0 and abs(scores[0, 1] - 1.0) < 1e-4       # a perturbed / identical copy ranks first
    assert all(sorted(ranks[:, j]) == list(range(5)) for j in range(2))
    assert np.all(np.diff(np.take_along_axis(scores, ranks, 0), axis=0) <= 0)              # best first
    meta, ranks, scores = rank_images({"network": copy.deepcopy(emb)}, (db,))
    assert ranks.shape == (5, 5) and list(ranks[0]) == list(range(5))                      # every image retrieves itself first
    with pytest.raises(AssertionError):
        validate({"network": emb, "data": {}}, (db,))
    with pytest.raises(NotImplementedError):                                               # a reference scenario's validation block
        validate({"network": copy.deepcopy(emb), "validation": {"type": "cirtorch", "dataset": "roxford5k"}, "data": {}}, (db,))
    with pytest.raises(NotImplementedError):
        validate({"network": copy.deepcopy(emb), "validation": {}, "data": {"test": {"dataset": "x"}}}, (db,))
    import gandtr_amd.learning as L
    net = L.load_network(copy.deepcopy(emb), "cpu")
    assert extract_vectors(net, [], "cpu").shape == (512, 0)
    # forward_list == the loop over forward, on the CPU too (there the model's forward_many is the plain loop); generators have no forward_many: plain loop
    xs = [d.unsqueeze(0) for d in db[:3]]
    for a, b in zip(net.forward_list(xs), [net(x) for x in xs]):
        assert torch.equal(a, b)
    # equal-size grouping (one forward per group of same-size images) == the reference's image-by-image loop, input order kept
    net.model.load_state_dict(synth.vgg16_state(0))
    loop = extract_vectors(net, db, "cpu", batched=False)
    grouped = extract_vectors(net, db, "cpu", batched=True, max_batch=2)
    assert grouped.shape == loop.shape == (512, 5) and float((grouped - loop).abs().max()) < 1e-6


def test_hub_networks_carry_a_device_transform():
    """`.transform_device`: device-side counterpart of `.transform` for decoded uint8 images (ingest row of SURVEY.md section 8f)"""
    import hubconf
    net = hubconf.gem_vgg16_cyclegan(pretrained=False, device="cpu")
    t = net.transform_device
    assert t.clahe_clip == 1.0 and t.clahe_grid == 8 and t.normalize and t.mean == [0.485, 0.456, 0.406]
    assert "apply_clahe:1.0" in repr(t)
    g = hubconf.cyclegan(pretrained=False, device="cpu").transform_device
    assert g.clahe_clip is None and g.mean == [0.5, 0.5, 0.5]
    from gandtr_amd.ingest import DeviceTransform
    import pytest
    with pytest.raises(KeyError):
        DeviceTransform("pil2np | random_crop:224 | totensor", [[0.5] * 3, [0.5] * 3])
    with pytest.raises(NotImplementedError):
        DeviceTransform("pil2np | apply_clahe:1.0:8:luv | totensor", [[0.5] * 3, [0.5] * 3])


def test_whitening_stage_signatures_without_a_device():
    """stage ABI of mdir/stages/whiten.py: (params, data) -> (metadata, *columns); the trivial branches need no device"""
    import numpy as np
    import pytest
    import torch
    import mdir.stages.whiten as stage
    from gandtr_amd.stages import FUNCTIONS
    assert FUNCTIONS["mdir.stages.whiten.learn_lw_whitening"] is stage.learn_lw_whitening
    assert stage.learn_lw_whitening({}, ([], np.zeros((0, 4)), [], [])) == ({"status": "Empty whitening produced"}, None)
    meta, names, values = stage.whiten({"dimensions": None}, (None, ["a"], np.zeros((1, 4))))
    assert meta == {"status": "No whitening applied"} and names == ["a"]
    with pytest.raises(AssertionError):
        stage.whiten({"dimensions": None, "bogus": 1}, (None, [], np.zeros((0, 4))))
    with pytest.raises(AssertionError):
        stage.learn_lw_whitening({}, (["a"], np.zeros((2, 4)), [], []))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            stage.learn_lw_whitening({}, (["a", "b"], np.zeros((2, 4)), ["a"], ["b"]))


def test_restricted_checkpoint_loader_names_the_offending_type(tmp_path):
    """a hub checkpoint carrying a type outside torch's allow-list must fail with a message that says so (and how to admit
    the type), not with a bare UnpicklingError from inside hubconf; a plain payload loads."""
    import pathlib
    from gandtr_amd.learning.network import load_restricted
    good, bad = tmp_path / "good.pth", tmp_path / "bad.pth"
    torch.save({"network_params": {"model": {"type": "x"}}, "model_state": {"w": torch.ones(2)}}, good)
    torch.save({"network_params": {"path": pathlib.PurePosixPath("/a")}}, bad)
    assert load_restricted(good)["model_state"]["w"].sum() == 2
    with pytest.raises(RuntimeError, match="add_safe_globals"):
        load_restricted(bad)


def test_precision_contract_and_env_knob(monkeypatch):
    """The precision contract of the HIP path is visible at the boundary (README first screen, hub/model.py docstrings): generators
    default to "f16c" (north_star's 1e-3 of the pre-tanh range), embedders and HED to "f16"; GANDTR_HIP_PRECISION overrides the family default on a
    hub object, a per-module ``hip_precision`` overrides both."""
    monkeypatch.delenv("GANDTR_HIP_PRECISION", raising=False)
    gen = hubconf.cyclegan(pretrained=False, device="cpu")
    emb = hubconf.gem_vgg16_cyclegan(pretrained=False, device="cpu")
    assert gen.model._hip_precision() == "f16c" and emb.model._hip_precision() == "f16"
    monkeypatch.setenv("GANDTR_HIP_PRECISION", "f16x3")
    assert gen.model._hip_precision() == "f16x3" and emb.model._hip_precision() == "f16x3"
    gen.model.hip_precision = "f16ch"
    assert gen.model._hip_precision() == "f16ch"
    monkeypatch.delenv("GANDTR_HIP_PRECISION")
    assert gen.model._hip_precision() == "f16ch" and emb.model._hip_precision() == "f16"
    import gandtr_amd.hub.model as hub_model
    for fn in (hub_model.cyclegan, hub_model.hedngan):
        assert "f16c" in fn.__doc__ and "1e-3" in fn.__doc__ and "f16x3" in fn.__doc__
