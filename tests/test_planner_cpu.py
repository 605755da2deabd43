"""Host logic of the C++ graph executor without a GPU: graph construction from reference state dicts, shape inference,
algorithmic FLOP counts (against the numbers the survey measured with forward hooks on the REFERENCE modules, BASELINE.md
section 3), workspace planning and argument validation.  Nothing here launches a kernel or touches HIP."""
import pytest

from gandtr_amd import engine
from gandtr_amd.tools import synth

DEV = "cuda:0"        # only recorded; no device call happens before finalize()


def _gflop(net, n, h, w):
    return net.flops(n, h, w) / 1e9


def test_generator_graph_flops_shapes_workspace():
    for norm in ("instance", "batch"):
        net = engine.build_generator(synth.generator_state(0, norm), DEV, precision="f16", finalize=False)
        assert _gflop(net, 1, 256, 256) == pytest.approx(99.10, abs=0.01)          # 99.10 GFLOP / 256^2 image
        assert _gflop(net, 4, 256, 256) == pytest.approx(4 * 99.10, abs=0.05)
        assert net.output_shapes(4, 256, 256) == [(4, 3, 256, 256)]
        assert net.output_shapes(1, 100, 76) == [(1, 3, 100, 76)]
        assert net.output_shapes(1, 101, 77) == [(1, 3, 104, 80)]                   # stride-2 floors, transposed convs double
        w1, w4 = net.workspace_bytes(1, 256, 256), net.workspace_bytes(4, 256, 256)
        assert 0 < w1 < w4 <= 4 * w1 + 4096
        # liveness-based reuse: far less than the sum of all activations (>= 60 fp16 tensors of up to 8 MiB per image)
        assert w1 < 40 * (1 << 20)
    exact = engine.build_generator(synth.generator_state(0, "instance"), DEV, precision="f16x3", finalize=False)
    assert _gflop(exact, 1, 256, 256) == pytest.approx(99.10, abs=0.01)            # algorithmic, not 3x
    assert exact.workspace_bytes(1, 256, 256) > net.workspace_bytes(1, 256, 256)    # fp32 activations
    comp = engine.build_generator(synth.generator_state(0, "instance"), DEV, finalize=False)      # the default: "f16c"
    assert comp.precision == "f16c"
    assert _gflop(comp, 1, 256, 256) == pytest.approx(99.10, abs=0.01)            # algorithmic, not 1.5x
    # fp32 activations as f16x3 (folding a norm with write-back keeps the raw and the normalised tensor alive together: a little
    # more scratch than f16x3 at batch 64, still a small fraction of the 288 GB)
    assert comp.workspace_bytes(1, 256, 256) > net.workspace_bytes(1, 256, 256)
    assert comp.workspace_bytes(64, 256, 256) < 4 * (1 << 30)


def test_generator_taps_shapes():
    sd = synth.generator_state(0, "instance", ngf=8, n_blocks=2)
    net = engine.build_generator(sd, DEV, taps=(1, 3, 6, 9, 11, 14, 17, 19), finalize=False)
    shapes = net.output_shapes(2, 32, 32)
    got = {t: shapes[s] for t, s in net.tap_slots.items()}
    assert got == {1: (2, 8, 32, 32), 3: (2, 8, 32, 32), 6: (2, 16, 16, 16), 9: (2, 32, 8, 8), 11: (2, 32, 8, 8),
                   14: (2, 16, 16, 16), 17: (2, 8, 32, 32), 19: (2, 3, 32, 32)}
    assert shapes[net.out_slot] == (2, 3, 32, 32)


def test_embedder_graph_flops_and_shapes():
    r101 = engine.build_embedder(synth.resnet101_state(0), DEV, feature_tap=True, finalize=False)
    assert _gflop(r101, 1, 1024, 1024) == pytest.approx(326.0, abs=0.1)
    assert _gflop(r101, 1, 724, 724) == pytest.approx(167.3, abs=0.1)
    assert _gflop(r101, 1, 512, 512) == pytest.approx(81.5, abs=0.1)
    assert _gflop(r101, 1, 256, 256) == pytest.approx(20.37, abs=0.02)
    shapes = r101.output_shapes(2, 1024, 1024)
    assert shapes[r101.out_slot] == (2, 2048) and shapes[r101.feature_slot] == (2, 2048, 32, 32)     # stride 32
    vgg = engine.build_embedder(synth.vgg16_state(0), DEV, feature_tap=True, finalize=False)
    assert _gflop(vgg, 1, 1024, 1024) == pytest.approx(641.4, abs=0.1)
    shapes = vgg.output_shapes(1, 1024, 1024)
    assert shapes[vgg.out_slot] == (1, 512) and shapes[vgg.feature_slot] == (1, 512, 64, 64)         # last max-pool dropped
    # F.interpolate(scale_factor) output sizes used by the multi-scale wrapper (1024 -> 724 / 512 / 1448)
    assert engine.HipNet.resized_size(1024, 1024, 2 ** -0.5) == (724, 724)
    assert engine.HipNet.resized_size(1024, 1024, 0.5) == (512, 512)
    assert engine.HipNet.resized_size(1024, 1024, 2 ** 0.5) == (1448, 1448)


def test_hed_graph_flops():
    hed = engine.build_hed(synth.hed_state(0), DEV, finalize=False)
    assert _gflop(hed, 1, 256, 256) == pytest.approx(40.1, abs=0.2)                # HED 40.1 GFLOP / 256^2
    assert hed.output_shapes(2, 64, 96) == [(2, 1, 64, 96)]


def test_planner_rejects_impossible_geometries():
    gen = engine.build_generator(synth.generator_state(0, "instance", ngf=8, n_blocks=1), DEV, finalize=False)
    with pytest.raises(ValueError):
        gen.workspace_bytes(1, 3, 3)                 # ReflectionPad2d(3) needs more than 3 pixels
    vgg = engine.build_embedder(synth.vgg16_state(0, width_div=4), DEV, finalize=False)
    with pytest.raises(ValueError):
        vgg.workspace_bytes(1, 8, 8)                 # four max-pools leave nothing
    with pytest.raises(ValueError):
        gen.workspace_bytes(0, 32, 32)


def test_builder_validation():
    net = engine.HipNet(DEV)
    t = net.input(3)
    with pytest.raises(ValueError):
        net.conv(t, synth._normal(0, "w", (16, 5, 3, 3)))          # cin 5 pads to 8 == ok; but wrong tensor? (3 -> 8 ok)
    net2 = engine.HipNet(DEV)
    t2 = net2.input(3)
    a = net2.conv(t2, synth._normal(0, "w", (16, 3, 3, 3)), pad=1)
    with pytest.raises(ValueError):
        net2.conv(a, synth._normal(0, "w", (16, 32, 3, 3)), pad=1)  # channel mismatch 16 vs 32
    with pytest.raises(ValueError):
        net2.conv(a, synth._normal(0, "w", (12, 16, 3, 3)), pad=1)  # internal cout must be a multiple of 8
    with pytest.raises(ValueError):
        net2.conv(a, synth._normal(0, "w", (16, 16, 5, 5)), stride=2, pad=2, transposed=True)   # only ConvT(k3,s2,p1,op1)
    with pytest.raises(ValueError):
        net2.instance_norm(a, residual=t2)                           # residual channels differ
    with pytest.raises(ValueError):
        net2.gem_l2n(a, 3.0)                                         # GeM needs channels % 64 == 0
    with pytest.raises(ValueError):
        engine.HipNet(DEV, precision="bf16")
    with pytest.raises(RuntimeError):
        net2.forward(None)                                           # not finalized


def test_planner_fusion_decisions_at_the_benchmarked_geometries(monkeypatch):
    """The fusion decisions the planner takes (gdt_net_plan_summary: pure host logic) at the geometries the numbers are quoted on -- a regression guard that runs without a
    GPU: a change of an eligibility rule that silently un-fuses a layer shows up here, not only as a slower bench line."""
    for k in ("GDT_CONV_XEXP", "GDT_XEXP_CHAIN", "GDT_CONV_BNECK", "GDT_XEXP_PH", "GDT_NORM_FUSION", "GDT_CONV_1X1_CAT"):
        monkeypatch.delenv(k, raising=False)
    r101 = engine.build_embedder(synth.resnet101_state(0), DEV, finalize=False)
    p = r101.plan_summary(32, 1024, 1024)
    # torchvision ResNet-101 [3, 4, 23, 3] (imageretrievalnet.py:189-190): layer1's three blocks (one with a projection) and layer2's three identity blocks run as one
    # launch each; layer3's 22 identity blocks as 3x3 + expand launches, 21 of them with the next block's reduce conv chained in; the projection shortcuts of the first
    # blocks of layer2 / 3 / 4 ride in their expand convs; the stem reads the fp32 image itself and writes the pooled tensor
    assert p["bottlenecks_fused"] == 6 and p["conv3x3_expand"] == 22 and p["chained_reduce"] == 21 and p["shortcuts_folded"] == 3, p
    assert p["direct_stem"] == 1 and p["pools_fused"] == 1 and p["norms_folded"] == 0, p
    assert p["conv_launches"] == 104 - 2 * 6 - 1 - 22 - 21 - 3, p             # 104 convs (+ the projection of layer1's first block inside its Bottleneck launch)
    assert r101.plan_summary(32, 1024, 1024, resize=True)["direct_stem"] == 0  # a resized pyramid level packs its input first
    small = r101.plan_summary(8, 512, 512)                                     # the small level of config 4's pyramid: too few patches for the layer3 fusion
    assert small["conv3x3_expand"] == 0 and small["chained_reduce"] == 0, small
    # ... unless it runs CONCURRENTLY with the other levels (the planner's group hint, gdt_net_set_group_factor): 8 x 1024^2 has 128 patches of 16 x 16 in layer3,
    # the hub-default pyramid 128 + 72 + 32 -- together they fill the chip, and the fused launches win (measured: 7.40 -> 6.72 ms); below ~200 in the group they lose
    eight = r101.plan_summary(8, 1024, 1024)
    assert eight["conv3x3_expand"] == 0, eight
    r101.set_group_factor(1.75)
    assert r101.plan_summary(8, 1024, 1024)["conv3x3_expand"] == 22 and r101.plan_summary(4, 1024, 1024)["conv3x3_expand"] == 0
    assert r101.workspace_bytes(8, 1024, 1024) > 0
    r101.set_group_factor(1.0)
    assert r101.plan_summary(8, 1024, 1024) == eight
    monkeypatch.setenv("GDT_XEXP_CHAIN", "0")
    assert r101.plan_summary(32, 1024, 1024)["chained_reduce"] == 0
    monkeypatch.delenv("GDT_XEXP_CHAIN")
    vgg = engine.build_embedder(synth.vgg16_state(0), DEV, finalize=False)
    assert vgg.plan_summary(32, 1024, 1024)["pools_fused"] == 4                # MaxPool2d(2, 2) x 4 in the producing convs' epilogues
    gen = engine.build_generator(synth.generator_state(0, "instance"), DEV, finalize=False)        # default precision f16c
    g = gen.plan_summary(64, 256, 256)
    # p2p_networks.py:269-311: all 23 InstanceNorms are applied by their consumers' staging; both transposed convs as one fused-phase launch, both stride-2 convs as shift forms
    assert g["norms_folded"] == 23 and g["transposed_fused"] == 2 and g["stride2_shift"] == 2 and g["conv_launches"] == 24, g
    g1 = gen.plan_summary(1, 64, 64)
    assert g1["norms_folded"] < 23, g1                                         # a single small image: the generic kernels, own normalisation passes
    # the exact mode ("f16x3") at the benchmarked batch: the stride-2 convs as shift forms on the patch kernel, 22 of the 23 norms folded (resblock norms with their
    # residual and write-back; the last resblock's feeds the phase launches of a transposed conv and keeps its own pass)
    for k in ("GDT_CONV_HALO_X3", "GDT_CONV_HALO_X3_FORMS", "GDT_X3_NORM_FOLD", "GDT_HEAD7_X3"):
        monkeypatch.delenv(k, raising=False)
    exact = engine.build_generator(synth.generator_state(0, "instance"), DEV, precision="f16x3", finalize=False)
    e = exact.plan_summary(64, 256, 256)
    assert e["norms_folded"] == 22 and e["stride2_shift"] == 2 and e["transposed_fused"] == 0, e
    monkeypatch.setenv("GDT_X3_NORM_FOLD", "1")                                # the first form: plain norm + ReLU into the stride-1 patch kernel only
    assert exact.plan_summary(64, 256, 256)["norms_folded"] == 10                # (9 inside the resblocks + the head's)
    monkeypatch.delenv("GDT_X3_NORM_FOLD")
