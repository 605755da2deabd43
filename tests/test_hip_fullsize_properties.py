"""BASELINE.json's full-size configurations on the GPU, checked through size-independent properties (the CPU oracle would
need minutes per case at these sizes): batch independence (an image's result does not depend on its batch mates, bitwise),
run-to-run determinism (fixed-order reductions, no atomics), shard-concatenation == full batch (the multi-GPU contract of
SURVEY.md section 8e, emulated on one GPU), unit-norm descriptors, identities of the aggregation / whitening ops."""
import pytest
import torch

from gandtr_amd import engine
from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f16c", "f16", "f16x3"])
def test_generator_64x256_batch_independence_and_determinism(cuda_device, precision):
    """config 2 geometry (64x3x256x256).  Tile shapes and kernel variants change with the batch size (halo kernel vs generic
    implicit GEMM), so cross-batch equality is numerical, not bitwise; same-geometry runs must be bitwise identical."""
    net = engine.build_generator(synth.generator_state(0, "instance"), cuda_device, pre_tanh=True, precision=precision)
    n = 16 if precision == "f16x3" else 64
    x = synth.synth_input(40, (n, 3, 256, 256), 1.0).to(cuda_device)
    full = net.forward(x)[net.out_slot]
    again = net.forward(x)[net.out_slot]
    assert torch.equal(full, again)                                   # deterministic
    assert torch.isfinite(full).all() and full.shape == (n, 3, 256, 256)
    half = torch.cat([net.forward(x[: n // 2])[net.out_slot], net.forward(x[n // 2:])[net.out_slot]])
    if precision != "f16x3":
        assert torch.equal(half, full)                                # same kernel variants: shard concat == full batch, bitwise
    else:                                                             # 8 images fall below the halo kernel's tile threshold
        assert float((half - full).abs().max() / full.abs().max()) < 2e-5
    single = net.forward(x[5:6])[net.out_slot]
    tol = {"f16": 6e-3, "f16c": 1e-3, "f16x3": 2e-5}[precision]     # different kernel variants at batch 1 (f16c: generic f16x3 kernels there)
    assert float((single - full[5:6]).abs().max() / full.abs().max()) < tol
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0)).to(cuda_device)
    assert torch.equal(net.forward(x[perm])[net.out_slot], full[perm])   # images are independent units
    # the production geometry itself (batch 64: persistent halo / transposed / stem / head kernels, every InstanceNorm folded)
    # against the CPU oracle on two of its images -- same gates as test_hip_models.py: north_star's 1e-3 in the default f16c mode and
    # in f16x3, the measured fp16 envelope in the opt-in single-pass mode
    from oracle import gandtr_oracle as O
    sd = synth.generator_state(0, "instance")
    for i in (5, n - 1):
        ref = O.resnet_generator(x[i:i + 1].cpu(), sd, "instance", 9, pre_tanh=True)
        err = float((full[i:i + 1].cpu() - ref).abs().max() / ref.abs().max())
        assert err < (3.5e-3 if precision == "f16" else 1e-3), (i, err)


def test_hedngan_bn_generator_64x256_against_oracle(cuda_device):
    """BASELINE config 3's generator as the reference defines it (hub/model.py:139-154: norm_layer = batch, no conv bias) at the BENCHMARKED
    geometry: batch 64 selects the 256-column form of the compensated patch kernel, whose BatchNorm variant adds the ResnetBlock residual
    in the pipelined epilogue (p2p_networks.py:505 with the eval-mode norm folded into the conv).  Batches of 2 / 8 (test_hip_models.py)
    run the 128-column form, so this is the only oracle comparison of the variant the config-3 number is measured on: two images of the
    64 against O.resnet_generator, north_star's 1e-3 of the pre-tanh range, default precision (f16c)."""
    from oracle import gandtr_oracle as O
    sd = synth.generator_state(0, "batch")
    net = engine.build_generator(sd, cuda_device, pre_tanh=True)
    assert net.precision == "f16c"
    x = synth.synth_input(43, (64, 3, 256, 256), 1.0).to(cuda_device)
    net.set_profiling(True)
    full = net.forward(x)[net.out_slot]
    torch.cuda.synchronize()
    variants = [v for kind, v, _, fl in net.profile() if kind == 1 and fl > 0]
    net.set_profiling(False)
    assert variants.count(970256) + variants.count(971256) == 18, variants      # every resblock conv ran the 256-column compensated patch kernel (971256: its 16 x 16 MFMA form)
    assert torch.equal(full, net.forward(x)[net.out_slot])
    for i in (0, 37):
        ref = O.resnet_generator(x[i:i + 1].cpu(), sd, "batch", 9, pre_tanh=True)
        err = float((full[i:i + 1].cpu() - ref).abs().max() / ref.abs().max())
        assert err < 1e-3, (i, err)


def test_hedngan_64x256_with_hed(cuda_device):
    gen = engine.build_generator(synth.generator_state(0, "batch"), cuda_device)
    hed = engine.build_hed(synth.hed_state(0), cuda_device, perm=[2, 1, 0], in_affine=([0.5] * 3, [0.09212946, 0.04247542, 0.01890622]))
    x = synth.synth_input(41, (64, 3, 256, 256), 1.0).to(cuda_device)
    y = gen.forward(x)[gen.out_slot]
    e = hed.forward(y)[hed.out_slot]
    assert y.shape == (64, 3, 256, 256) and float(y.abs().max()) <= 1.0
    assert e.shape == (64, 1, 256, 256) and float(e.min()) >= 0.0 and float(e.max()) <= 1.0
    assert torch.equal(e[:8], hed.forward(gen.forward(x)[gen.out_slot])[hed.out_slot][:8])
    # BASELINE config 3's HED leg AT ITS OWN GEOMETRY against the oracle (round-4 verdict: the fixture comparison runs at 2 x 64 x 96, which selects other
    # forms of conv3x3_halo_rb -- no four-wave 64 / 128-channel forms, no fused pools -- than 64 x 256 x 256): two of the 64 images, the device's own
    # generator output as the input of both sides (edges_epochs.py:87 forward-only: rgb2bgr_pre, meanstd_pre, HedInterpolation.forward hed.py:60-83)
    from oracle import gandtr_oracle as O
    hed_sd = synth.hed_state(0)
    for i in (5, 63):
        ref = O.hed_on_generator_output(y[i:i + 1].cpu(), hed_sd)
        err = float((e[i:i + 1].cpu() - ref).abs().max())
        print("HED @64x256x256, image %d: max|d| %.2e (edge map in [0, 1], ref range %.3f..%.3f)" % (i, err, float(ref.min()), float(ref.max())))
        assert err <= 1e-3, (i, err)


@pytest.mark.parametrize("arch,n", [("resnet101", 8), ("vgg16", 4)])
def test_embedder_1024_properties(cuda_device, arch, n):
    sd = synth.resnet101_state(0) if arch == "resnet101" else synth.vgg16_state(0)
    net = engine.build_embedder(sd, cuda_device)
    x = synth.synth_input(42, (n, 3, 1024, 1024)).to(cuda_device)
    d = net.forward(x)[net.out_slot]                                   # N x D
    assert d.shape == (n, 2048 if arch == "resnet101" else 512)
    assert torch.allclose(d.norm(dim=1), torch.ones(n, device=cuda_device), atol=1e-5)
    assert torch.equal(d, net.forward(x)[net.out_slot])
    shards = torch.cat([net.forward(x[: n // 2])[net.out_slot], net.forward(x[n // 2:])[net.out_slot]])
    cos = torch.nn.functional.cosine_similarity(shards, d, dim=1)
    assert float(cos.min()) > 0.99999                                  # rank shards concatenated == whole batch
    # the full-size geometry (patch kernels, fused max pools, stem kernel) against the CPU oracle on one image: north_star gates
    from oracle import gandtr_oracle as O
    ref = O.image_retrieval_forward(x[1:2].cpu(), sd, arch).t().contiguous()[0]
    got = d[1].cpu()
    assert float(torch.nn.functional.cosine_similarity(got, ref, dim=0)) >= 0.9999
    assert float((got - ref).abs().max()) <= 1e-3
    # hub-default pyramid {1, 1/sqrt2, 1/2} -> 1024, 724, 512 inside the pack kernel; aggregate; whiten with the identity
    scales = [1.0, 2 ** -0.5, 0.5]
    per_scale = torch.stack([net.forward(x[:2], scale=s)[net.out_slot] for s in scales])
    v = engine.ms_aggregate(per_scale, float(sd["pool.p"]))
    assert torch.allclose(v.norm(dim=1), torch.ones(2, device=cuda_device), atol=1e-5)
    D = v.shape[1]
    w = engine.whiten(v, torch.eye(D, device=cuda_device), torch.zeros(D, device=cuda_device))
    assert torch.allclose(w, v / (v.norm(dim=1, keepdim=True) + 1e-6), atol=1e-6)


@pytest.mark.parametrize("arch", ["resnet101", "vgg16"])
def test_embedder_bench_geometry_against_oracle(cuda_device, arch):
    """The geometries bench.py / bench_configs.py quote numbers on -- GeM-ResNet-101 and GeM-VGG16 at 32 x 3 x 1024 x 1024 (BASELINE
    configs 2 and the secondary headline) -- against the CPU oracle on two of the 32 images.  Kernel choice depends on the tile count
    (conv3x3_halo_rb / conv1x1_rb thresholds), so batch 32 runs variants that batch 4 / 8 do not: north_star's gates hold HERE:
    cosine >= 0.9999, ||d||inf <= 1e-3."""
    from oracle import gandtr_oracle as O
    sd = synth.resnet101_state(0) if arch == "resnet101" else synth.vgg16_state(0)
    net = engine.build_embedder(sd, cuda_device)
    x = synth.synth_input(43, (32, 3, 1024, 1024))
    d = net.forward(x.to(cuda_device))[net.out_slot].cpu()                                    # N x D
    assert d.shape == (32, 2048 if arch == "resnet101" else 512)
    for i in (3, 31):
        ref = O.image_retrieval_forward(x[i:i + 1], sd, arch).t().contiguous()[0]
        cos = float(torch.nn.functional.cosine_similarity(d[i], ref, dim=0))
        err = float((d[i] - ref).abs().max())
        print("%s batch 32 @1024, image %d: cos %.7f, |d|inf %.2e" % (arch, i, cos, err))
        assert cos >= 0.9999 and err <= 1e-3, (i, cos, err)


def test_resnet101_sms_pyramid_at_1024_against_oracle(cuda_device):
    """BASELINE config 4's `sms` preset {1, 1/sqrt2, sqrt2} on a 1 x 3 x 1024 x 1024 image: the 1448 x 1448 level (bilinear
    up-sampling inside the input-pack kernel, the largest tensors of any config) + aggregation + learned whitening against
    O.embed_ms_whiten (mdir/components/data/wrapper.py:197-263, 308-322): cosine >= 0.9999, ||d||inf <= 1e-3."""
    from oracle import gandtr_oracle as O
    sd = synth.resnet101_state(0, p=3.0)
    lw = synth.whitening_state(0, 2048)
    P, m = torch.from_numpy(lw["P"]), torch.from_numpy(lw["m"])
    x = synth.synth_input(44, (1, 3, 1024, 1024))
    scales = O.SCALE_PRESETS["sms"]
    ref = O.embed_ms_whiten(x, sd, "resnet101", scales, P, m).reshape(-1)
    net = engine.build_embedder(sd, cuda_device)
    xd = x.to(cuda_device)
    per_scale = torch.stack([net.forward(xd, scale=s)[net.out_slot] for s in scales])       # S x 1 x D
    v = engine.ms_aggregate(per_scale, 3.0)
    got = engine.whiten(v, P.to(cuda_device), m.to(cuda_device)).cpu().reshape(-1)
    cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
    err = float((got - ref).abs().max())
    print("resnet101 sms @1024 (1448 level): cos %.7f, |d|inf %.2e" % (cos, err))
    assert cos >= 0.9999 and err <= 1e-3, (cos, err)


def test_resnet101_hub_default_pyramid_at_1024_against_oracle(cuda_device):
    """The twin of the `sms` test for the drop-in DEFAULT scales {1, 1/sqrt2, 1/2} (embedding.yml:25, wrapper.py:207-208): a 1 x 3 x 1024 x 1024 image
    -> levels 1024 / 724 / 512 (the 512 level of a 1024 image was oracle-checked at small sizes only) + aggregation + learned whitening against
    O.embed_ms_whiten -- through `forward_many`, the path the hub network takes for a pyramid: cosine >= 0.9999, ||d||inf <= 1e-3."""
    from oracle import gandtr_oracle as O
    sd = synth.resnet101_state(0, p=3.0)
    lw = synth.whitening_state(0, 2048)
    P, m = torch.from_numpy(lw["P"]), torch.from_numpy(lw["m"])
    x = synth.synth_input(45, (1, 3, 1024, 1024))
    scales = O.SCALE_PRESETS[True]
    ref = O.embed_ms_whiten(x, sd, "resnet101", scales, P, m).reshape(-1)
    net = engine.build_embedder(sd, cuda_device)
    xd = x.to(cuda_device)
    per_scale = torch.stack([o[net.out_slot] for o in net.forward_many([(xd, s) for s in scales])])       # S x 1 x D
    one_by_one = torch.stack([net.forward(xd, scale=s)[net.out_slot] for s in scales])
    assert torch.equal(per_scale, one_by_one)
    v = engine.ms_aggregate(per_scale, 3.0)
    got = engine.whiten(v, P.to(cuda_device), m.to(cuda_device)).cpu().reshape(-1)
    cos = float(torch.nn.functional.cosine_similarity(got, ref, dim=0))
    err = float((got - ref).abs().max())
    print("resnet101 hub-default pyramid @1024 (724 / 512 levels): cos %.7f, |d|inf %.2e" % (cos, err))
    assert cos >= 0.9999 and err <= 1e-3, (cos, err)


def test_config4_per_rank_workload_against_oracle(cuda_device):
    """BASELINE config 4's per-rank workload as it runs since round 5: 8 x 3 x 1024 x 1024, hub-default pyramid, the three levels concurrently on side streams WITH the
    planner's group hint -- the layer3 fusion (3x3 + expand + the next block's reduce conv in one launch) is then chosen for levels that would not fuse alone (128 + 72 +
    32 patches fill the chip together).  Two of the eight images against O.embed_ms_whiten: cosine >= 0.9999, ||d||inf <= 1e-3; and the plan the hint produces."""
    from oracle import gandtr_oracle as O
    sd = synth.resnet101_state(0, p=3.0)
    lw = synth.whitening_state(0, 2048)
    P, m = torch.from_numpy(lw["P"]), torch.from_numpy(lw["m"])
    x = synth.synth_input(46, (8, 3, 1024, 1024))
    scales = O.SCALE_PRESETS[True]
    net = engine.build_embedder(sd, cuda_device)
    alone = net.plan_summary(8, 1024, 1024)
    net.set_group_factor(1.75)
    grouped = net.plan_summary(8, 1024, 1024)
    net.set_group_factor(1.0)
    assert alone["conv3x3_expand"] == 0 and grouped["conv3x3_expand"] == 22 and grouped["chained_reduce"] == 21, (alone, grouped)
    xd = x.to(cuda_device)
    per_scale = torch.stack([o[net.out_slot] for o in net.forward_many([(xd, s) for s in scales])])       # S x 8 x D
    assert net.plan_summary(8, 1024, 1024) == alone                                                          # (the hint is reset after the call)
    got = engine.whiten(engine.ms_aggregate(per_scale, 3.0), P.to(cuda_device), m.to(cuda_device)).cpu()
    for i in (0, 7):
        ref = O.embed_ms_whiten(x[i:i + 1], sd, "resnet101", scales, P, m).reshape(-1)
        cos = float(torch.nn.functional.cosine_similarity(got[i], ref, dim=0))
        err = float((got[i] - ref).abs().max())
        print("config 4 per-rank workload (8 x 1024^2, grouped plan), image %d: cos %.7f, |d|inf %.2e" % (i, cos, err))
        assert cos >= 0.9999 and err <= 1e-3, (i, cos, err)


def test_descriptor_op_identities(cuda_device):
    v = torch.rand(3, 5, 256, device=cuda_device) + 0.1
    same = v[:1].expand(3, 5, 256).contiguous()
    agg = engine.ms_aggregate(same, 3.0)                                # aggregating identical vectors == normalising one
    assert torch.allclose(agg, same[0] / same[0].norm(dim=1, keepdim=True), atol=1e-6)
    assert torch.allclose(engine.l2n_rows(v[0]), v[0] / (v[0].norm(dim=1, keepdim=True) + 1e-6), atol=1e-7)
    P = torch.randn(256, 256, device=cuda_device)
    m = torch.randn(256, device=cuda_device) * 0.1
    full = engine.whiten(v[0], P, m)
    assert torch.allclose(engine.whiten(v[0], P, m, 64), torch.nn.functional.normalize(
        (v[0] - m) @ P[:64].t(), dim=1), atol=1e-5)                     # dimensionality reduction = leading rows of P
    assert full.shape == (5, 256)
