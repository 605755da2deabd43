"""GPU parity of the "f16c" precision mode (fp16 MFMA product + block-scaled fp4 x fp6 correction product, conv3x3_halo_c.hip) layer by
layer, of the stand-alone GeM / L2N entry point against the reference's golden vectors, and of the validate-shaped caller.

A compensated conv is held to 2e-4 of an fp64 evaluation of the same layer on the same fp32 inputs -- single-pass fp16 measures
5e-4 on these layers, so a correction product that silently did nothing (wrong operand layout, wrong block scale) fails here."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gandtr_amd import engine
from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _g(seed, name, shape, std=1.0):
    return synth._normal(seed, name, shape, std)


@pytest.mark.parametrize("n", [8, 16])
@pytest.mark.parametrize("norm,res,reflect", [(False, False, True), (False, False, False), (True, False, True), (True, True, True)])
def test_halo_c_conv3x3(cuda_device, norm, res, reflect, n):
    """3x3 / stride 1 conv 256 -> 256 on n x 64 x 64: plain, with the producer's InstanceNorm + ReLU folded into the staging (+ write-back:
    the normalised tensor is also an output), and with the ResnetBlock form x + IN(conv) folded in.  The statistics of ITS output feed a
    following InstanceNorm.  n = 8: 128 patches, the 128-column form of the persistent compensated kernel; n = 16: 256 patches, the
    256-column form the batch-64 benchmark runs (gdt_conv_halo_c_columns)."""
    cin = cout = 256
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    t0 = net.conv(t, _g(0, "w0", (cin, 3, 1, 1), 0.7))
    t = t0
    if norm:
        r = net.conv(t0, _g(0, "w1", (cin, cin, 1, 1), 0.06)) if res else -1
        t = net.instance_norm(t0, relu=not res, residual=r)
    wt, bias = _g(0, "w", (cout, cin, 3, 3), 0.05), _g(0, "b", (cout,), 0.2)
    out = net.conv(t, wt, bias, pad=1, reflect=reflect)
    o2 = net.instance_norm(out, relu=True)
    taps = [net.output_nchw(out), net.output_nchw(o2)] + ([net.output_nchw(t)] if norm else [])
    net.finalize()
    x = synth.synth_input(1, (n, 3, 64, 64))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    ran = [v for k, v, ms, fl in net.profile() if k == 1]
    assert (970128 in ran) if n == 8 else (971256 in ran or 970256 in ran)      # (971256: conv3x3_halo_c16.hip, GDT_CONV_HALO_C16=0 switches it off)
    a0 = F.conv2d(x.double(), _g(0, "w0", (cin, 3, 1, 1), 0.7).double())
    a = a0
    if norm:
        a = F.instance_norm(a0, eps=1e-5)
        a = a + F.conv2d(a0, _g(0, "w1", (cin, cin, 1, 1), 0.06).double()) if res else F.relu(a)
    ai = F.pad(a, (1,) * 4, mode="reflect") if reflect else a
    ref = F.conv2d(ai, wt.double(), bias.double(), padding=0 if reflect else 1)
    assert _rel(outs[taps[0]].double().cpu(), ref) < 2e-4
    assert _rel(outs[taps[1]].double().cpu(), F.relu(F.instance_norm(ref, eps=1e-5))) < 3e-4
    if norm:
        assert _rel(outs[taps[2]].double().cpu(), a) < 1e-5          # the write-back of the folded normalisation is plain fp32


@pytest.mark.parametrize("n", [8, 16])
def test_halo_c_conv3x3_epilogue_residual(cuda_device, n):
    """the BatchNorm generator's second ResnetBlock conv (hub hedngan: p2p_networks.py:503-506 with the eval-mode norm folded into the
    conv): y = x + BN(conv(pad(x))) -- the residual is added in the kernel's EPILOGUE (fetched one block ahead in the pipelined body),
    128-column form (n = 8) and the 256-column form of the batch-64 benchmark (n = 16), against fp64."""
    c = 256
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    t0 = net.conv(t, _g(0, "w0", (c, 3, 1, 1), 0.7))
    wt = _g(0, "w", (c, c, 3, 3), 0.05)
    bn = (1.0 + _g(0, "g", (c,), 0.2), _g(0, "be", (c,), 0.2), _g(0, "m", (c,), 0.3), 0.5 + _g(0, "v", (c,), 0.1).abs())
    out = net.conv(t0, wt, None, bn=bn, pad=1, reflect=True, residual=t0)
    tap = net.output_nchw(out)
    net.finalize()
    x = synth.synth_input(2, (n, 3, 64, 64))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    ran = [v for k, v, ms, fl in net.profile() if k == 1]
    assert (970128 in ran) if n == 8 else (971256 in ran or 970256 in ran)
    a0 = F.conv2d(x.double(), _g(0, "w0", (c, 3, 1, 1), 0.7).double())
    y = F.conv2d(F.pad(a0, (1,) * 4, mode="reflect"), wt.double())
    g, be, m, v = (q.double() for q in bn)
    ref = a0 + F.batch_norm(y, m, v, g, be, False, 0.0, 1e-5)
    assert _rel(outs[tap].double().cpu(), ref) < 2e-4


@pytest.mark.parametrize("kind", ["conv3x3", "stride2", "stride2_128", "transposed"])
def test_halo_c_ragged_patches(cuda_device, kind):
    """image sizes that are not multiples of the 16 x 16 patch (the checked epilogue body, halo rows beyond the image) on the
    compensated kernels: 3x3 (1 x 4 wave layout), stride 2 (256- and 128-column tiles), transposed; no statistics consumer
    (fused statistics need whole patches)"""
    n, h, w = 6, 76, 92
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    if kind == "conv3x3":
        cin, cout = 256, 256
        t0 = net.conv(t, _g(0, "w0", (cin, 3, 1, 1), 0.7))
        wt, bias = _g(0, "w", (cout, cin, 3, 3), 0.05), _g(0, "b", (cout,), 0.2)
        out = net.conv(t0, wt, bias, pad=1, reflect=False, relu=True)
        ref_fn = lambda a: F.relu(F.conv2d(a, wt.double(), bias.double(), padding=1))
    elif kind.startswith("stride2"):
        cin, cout = (64, 128) if kind == "stride2_128" else (128, 256)
        h, w = 2 * h, 2 * w                                         # ragged OUTPUT grid 76 x 92
        t0 = net.conv(t, _g(0, "w0", (cin, 3, 1, 1), 0.7))
        wt, bias = _g(0, "w", (cout, cin, 3, 3), 0.05), _g(0, "b", (cout,), 0.2)
        out = net.conv(t0, wt, bias, stride=2, pad=1)
        ref_fn = lambda a: F.conv2d(a, wt.double(), bias.double(), stride=2, padding=1)
    else:
        cin, cout = 256, 128
        t0 = net.conv(t, _g(0, "w0", (cin, 3, 1, 1), 0.7))
        wt, bias = _g(0, "w", (cin, cout, 3, 3), 0.05), _g(0, "b", (cout,), 0.2)
        out = net.conv(t0, wt, bias, stride=2, pad=1, transposed=True)
        ref_fn = lambda a: F.conv_transpose2d(a, wt.double(), bias.double(), stride=2, padding=1, output_padding=1)
    tap = net.output_nchw(out)
    net.finalize()
    x = synth.synth_input(5, (n, 3, h, w))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    variant = [v for k, v, ms, fl in net.profile() if k == 1][-1]
    assert variant in (970256, 970128, 980256, 990256), variant               # a compensated patch kernel ran the layer
    a0 = F.conv2d(x.double(), _g(0, "w0", (cin, 3, 1, 1), 0.7).double())
    ref = ref_fn(a0)
    got = outs[tap].double().cpu()
    assert got.shape == ref.shape
    assert _rel(got, ref) < 2e-4


@pytest.mark.parametrize("cin,cout,norm,res", [(256, 128, False, False), (256, 128, True, True), (128, 64, True, False)])
def test_halo_c_transposed(cuda_device, cin, cout, norm, res):
    """ConvTranspose2d(k3, s2, p1, op1) on the compensated kernel's transposed form (four input shifts, sub-pixel scatter), with the
    producer's InstanceNorm (+ReLU | + residual) folded in and the statistics of its output taken in the epilogue"""
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    t0 = net.conv(t, _g(0, "w0", (cin, 3, 1, 1), 0.7))
    t = t0
    if norm:
        r = net.conv(t0, _g(0, "w1", (cin, cin, 1, 1), 0.06)) if res else -1
        t = net.instance_norm(t0, relu=not res, residual=r)
    wt, bias = _g(0, "w", (cin, cout, 3, 3), 0.05), _g(0, "b", (cout,), 0.2)
    out = net.conv(t, wt, bias, stride=2, pad=1, transposed=True)
    o2 = net.instance_norm(out, relu=True)
    taps = [net.output_nchw(out), net.output_nchw(o2)]
    net.finalize()
    x = synth.synth_input(1, (8, 3, 64, 64))
    outs = net.forward(x.to(cuda_device))
    a0 = F.conv2d(x.double(), _g(0, "w0", (cin, 3, 1, 1), 0.7).double())
    a = a0
    if norm:
        a = F.instance_norm(a0, eps=1e-5)
        a = a + F.conv2d(a0, _g(0, "w1", (cin, cin, 1, 1), 0.06).double()) if res else F.relu(a)
    ref = F.conv_transpose2d(a, wt.double(), bias.double(), stride=2, padding=1, output_padding=1)
    assert outs[taps[0]].shape == ref.shape
    assert _rel(outs[taps[0]].double().cpu(), ref) < 3e-4
    assert _rel(outs[taps[1]].double().cpu(), F.relu(F.instance_norm(ref, eps=1e-5))) < 4e-4


@pytest.mark.parametrize("cin,cout,norm,tapped,bn", [(64, 128, False, False, False), (64, 128, True, False, False), (128, 256, True, True, False),
                                                      (64, 128, False, False, True)])
def test_halo_c_stride2(cuda_device, cin, cout, norm, tapped, bn):
    """Conv2d(k3, s2, p1, zero padding) -- the generator's two down-sampling layers (p2p_networks.py:274-280) -- on the compensated
    kernel's stride-2 form (2x2-shift conv over the virtual space-to-depth input), with the producer's InstanceNorm + ReLU folded in
    (and written back when the normalised tensor has another consumer), statistics of its own output, or folded BatchNorm + ReLU"""
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    t0 = net.conv(t, _g(0, "w0", (cin, 3, 1, 1), 0.7))
    t = net.instance_norm(t0, relu=True) if norm else t0
    wt = _g(0, "w", (cout, cin, 3, 3), 0.05)
    bias = None if bn else _g(0, "b", (cout,), 0.2)
    bnp = None
    if bn:
        bnp = (synth._uniform(0, "g", (cout,), 0.5, 1.5), _g(0, "be", (cout,), 0.2), _g(0, "m", (cout,), 0.2), synth._uniform(0, "v", (cout,), 0.5, 1.5))
    out = net.conv(t, wt, bias, bn=bnp, stride=2, pad=1, relu=bn)
    taps = [net.output_nchw(out)]
    if not bn:
        taps.append(net.output_nchw(net.instance_norm(out, relu=True)))
    if tapped:
        taps.append(net.output_nchw(t))
    net.finalize()
    x = synth.synth_input(1, (8, 3, 128, 128))
    outs = net.forward(x.to(cuda_device))
    a = F.conv2d(x.double(), _g(0, "w0", (cin, 3, 1, 1), 0.7).double())
    if norm:
        a = F.relu(F.instance_norm(a, eps=1e-5))
    ref = F.conv2d(a, wt.double(), None if bn else bias.double(), stride=2, padding=1)
    if bn:
        ref = F.relu(F.batch_norm(ref, bnp[2].double(), bnp[3].double(), bnp[0].double(), bnp[1].double(), training=False, eps=1e-5))
    assert outs[taps[0]].shape == ref.shape == (8, cout, 64, 64)
    assert _rel(outs[taps[0]].double().cpu(), ref) < 2e-4
    if not bn:
        assert _rel(outs[taps[1]].double().cpu(), F.relu(F.instance_norm(ref, eps=1e-5))) < 3e-4
    if tapped:
        assert _rel(outs[taps[2]].double().cpu(), a) < 1e-5


@pytest.mark.parametrize("k,stride,reflect,bn", [(7, 1, True, False), (7, 1, True, True), (3, 1, False, True), (7, 2, False, True)])
def test_stem_c(cuda_device, k, stride, reflect, bn):
    """First layer (image -> 64 channels) in the f16c mode: the fp16 stem kernel on pixel words augmented with the activations' own
    rounding residuals + a second MFMA with the weight residuals = an fp32-class result (1e-5 of an fp64 evaluation; single-pass
    fp16 measures 2.5e-4), InstanceNorm statistics or folded BatchNorm + ReLU in the epilogue"""
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    w = _g(0, "w", (64, 3, k, k), 0.1)
    bnp = None
    if bn:
        bnp = (synth._uniform(0, "g", (64,), 0.5, 1.5), _g(0, "be", (64,), 0.2), _g(0, "m", (64,), 0.2), synth._uniform(0, "v", (64,), 0.5, 1.5))
    raw = net.conv(t, w, None, bn=bnp, stride=stride, pad=k // 2, reflect=reflect, relu=bn)
    slots = [net.output_nchw(raw)]
    if not bn:
        slots.append(net.output_nchw(net.instance_norm(raw, relu=True)))
    net.finalize()
    x = synth.synth_input(5, (4, 3, 256, 256), 1.0)
    outs = net.forward(x.to(cuda_device))
    xi = F.pad(x.double(), (k // 2,) * 4, mode="reflect") if reflect else x.double()
    ref = F.conv2d(xi, w.double(), stride=stride, padding=0 if reflect else k // 2)
    if bn:
        ref = F.relu(F.batch_norm(ref, bnp[2].double(), bnp[3].double(), bnp[0].double(), bnp[1].double(), training=False, eps=1e-5))
    assert outs[slots[0]].shape == ref.shape
    assert _rel(outs[slots[0]].double().cpu(), ref) < 1e-5
    if not bn:
        assert _rel(outs[slots[1]].double().cpu(), F.relu(F.instance_norm(ref, eps=1e-5))) < 2e-5


@pytest.mark.parametrize("norm,act", [(True, 0), (True, 1), (False, 0)])
def test_head7_f32_input(cuda_device, norm, act):
    """Generator head (ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh, p2p_networks.py:309-311) in the f16c mode: the fused head kernel
    reads the mode's fp32 tensor, folds the producer's InstanceNorm + ReLU and rounds to fp16 ONCE while staging; the product is a
    single fp16 pass (last layer: its 3e-4 is not amplified downstream) -- gate 6e-4 of the pre-activation range"""
    net = HipNet(cuda_device, "f16c")
    t = net.input(3)
    t0 = net.conv(t, _g(0, "w0", (64, 3, 1, 1), 0.7))
    t = net.instance_norm(t0, relu=True) if norm else t0
    w, b = _g(0, "w", (3, 64, 7, 7), 0.02), _g(0, "b", (3,), 0.1)
    slot = net.conv(t, w, b, pad=3, reflect=True, out_f32=True, act=act)
    net.finalize()
    x = synth.synth_input(7, (4, 3, 128, 160))
    got = net.forward(x.to(cuda_device))[slot].double().cpu()
    a = F.conv2d(x.double(), _g(0, "w0", (64, 3, 1, 1), 0.7).double())
    if norm:
        a = F.relu(F.instance_norm(a, eps=1e-5))
    pre = F.conv2d(F.pad(a, (3,) * 4, mode="reflect"), w.double(), b.double())
    ref = torch.tanh(pre) if act else pre
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 6e-4 * float(pre.abs().max())


def test_f16c_small_geometries_fall_back_to_the_exact_split(cuda_device):
    """below the patch kernels' tile threshold / channel counts the mode runs the generic f16x3 kernels: same answers as f16x3"""
    sd = synth.generator_state(0, "instance", ngf=16, n_blocks=3)
    x = synth.synth_input(11, (3, 3, 36, 132), 1.0).to(cuda_device)
    a = engine.build_generator(sd, cuda_device, pre_tanh=True, precision="f16c")
    b = engine.build_generator(sd, cuda_device, pre_tanh=True, precision="f16x3")
    assert torch.equal(a.forward(x)[a.out_slot], b.forward(x)[b.out_slot])


def test_gem_l2n_golden(cuda_device):
    """stand-alone gdt_gem_l2n against the outputs of the reference's own LF.gem / LF.l2n (tests/golden/gem_l2n.npz, made by
    tests/golden/make_golden.py): p in {3, 2.37}, inputs with exact zeros and negatives (the clamp path, functional.py:22)"""
    g = np.load(os.path.join(G, "gem_l2n.npz"))
    x = torch.from_numpy(g["x"]).to(cuda_device)
    assert float((x == 0).float().mean()) > 0 or float((x < 0).float().mean()) > 0
    for p in ("3.0", "2.37"):
        pooled, normed = engine.gem_l2n(x, float(p))
        assert pooled.shape == g["gem_p" + p].shape
        assert float((pooled.cpu() - torch.from_numpy(g["gem_p" + p])).abs().max()) < 1e-6
        assert float((normed.cpu() - torch.from_numpy(g["l2n_p" + p])).abs().max()) < 1e-6
    # a feature map of the embedder's real size: against torch
    f = torch.rand(2, 2048, 32, 32, device=cuda_device) - 0.2
    pooled, normed = engine.gem_l2n(f, 3.0)
    ref = F.avg_pool2d(f.clamp(min=1e-6).pow(3.0), (32, 32)).pow(1.0 / 3.0)
    assert float((pooled - ref).abs().max()) < 1e-5
    assert float((normed - ref / (ref.norm(dim=1, keepdim=True) + 1e-6)).abs().max()) < 1e-6


def test_validate_stage_on_gpu(cuda_device, tmp_path):
    """the validate stage's arithmetic on the device: descriptor extraction (HIP embedder, equal sizes batched) and device-side ranking agree with
    the reference's two numpy lines on the same descriptors (cirscore.py:71-73)"""
    import copy
    from gandtr_amd.stages.validate import rank_images, extract_vectors
    import gandtr_amd.learning as L
    emb = {"type": "SingleNetwork",
           "model": {"architecture": "cirnet", "cir_architecture": "vgg16", "local_whitening": False, "pooling": "gem",
                     "pretrained": False, "regional": False, "whitening": False},
           "initialize": False, "path": None,
           "runtime": {"wrappers": "cirfaketuplebatch",
                       "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    net = L.load_network(copy.deepcopy(emb), "cpu")
    net.model.load_state_dict(synth.vgg16_state(0))                 # seeded weights (the reference leaves them unseeded)
    ck = tmp_path / "vgg.pth"
    torch.save(net.state_dict()["net"], ck)
    params = {"network": {"path": str(ck), "runtime": copy.deepcopy(emb["runtime"])}}
    db = [synth.synth_input(40 + i, (3, 96, 128 - 16 * (i % 3))) for i in range(12)]
    qs = [db[7] + 0.01 * synth.synth_input(50, db[7].shape), db[2]]
    meta, ranks, scores = rank_images(copy.deepcopy(params), (db, qs))
    assert ranks.shape == (12, 2) and scores.shape == (12, 2) and ranks[0, 1] == 2 and abs(scores[2, 1] - 1.0) < 1e-4
    net = L.load_network(copy.deepcopy(params["network"]), cuda_device).eval()
    vecs = extract_vectors(net, db, cuda_device)
    assert vecs.is_cuda and vecs.shape == (512, 12)
    v, q = vecs.cpu().numpy(), extract_vectors(net, qs, cuda_device).cpu().numpy()
    ref_scores = np.dot(v.T, q)
    assert np.abs(scores - ref_scores).max() < 1e-5
    # the device ranking orders the same scores (ties between near-identical descriptors may permute: compare the sorted scores)
    assert np.abs(np.take_along_axis(ref_scores, ranks, 0) - np.sort(ref_scores, axis=0)[::-1]).max() < 1e-5
    assert all(sorted(ranks[:, j]) == list(range(12)) for j in range(2))


def test_extract_vectors_batches_equal_sizes(cuda_device, tmp_path):
    """extract_vectors groups images of equal size into one forward per group (batched result == stack of the per-image
    results; the reference's loader loop is batch 1, imageretrievalnet.py:312-339): 64 images of 3 sizes through the hub's multi-scale +
    whitening wrappers -- output order kept; within one kernel-variant class the columns are bit-identical whatever the batch mates and
    their order (reversed list == reversed columns); against the image-by-image loop, whose batch-1 launches pick other tile shapes
    (another fp32 summation order in front of the fp16 activation store), every column agrees to 5e-5 (measured 1.6e-5, cosine 1 - 1e-9)."""
    import copy
    import pickle
    import hubconf
    from gandtr_amd.learning.checkpoints import Checkpoints
    from gandtr_amd.stages.validate import extract_vectors
    import gandtr_amd.learning.network as NW
    base = hubconf.gem_vgg16_cyclegan(pretrained=False, device="cpu")
    base.model.load_state_dict(synth.vgg16_state(0))
    sd = base.state_dict()["net"]
    sd["network_params"]["runtime"]["data"] = {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]}
    ck, lw = str(tmp_path / "vgg.pth"), str(tmp_path / "lw.pkl")
    torch.save(sd, ck)
    with open(lw, "wb") as f:
        pickle.dump(synth.whitening_state(0, 512), f)
    runtime = {"wrappers": {"train": None, "eval": {"0_cirwhiten": {"whitening": lw, "dimensions": None}, "1_cirmultiscale": {"scales": True}}}}
    net = NW.initialize_network(None, cuda_device, Checkpoints.load_network(ck), runtime).eval()
    sizes = [(96, 128), (128, 96), (112, 112)]
    imgs = [synth.synth_input(200 + i, (3,) + sizes[i % 3]) for i in range(64)]
    loop = extract_vectors(net, imgs, cuda_device, batched=False)
    grouped = extract_vectors(net, imgs, cuda_device)                  # default on a HIP device: batched
    small = extract_vectors(net, imgs, cuda_device, max_batch=5)       # groups split into several forwards
    assert loop.shape == grouped.shape == small.shape == (512, 64) and grouped.is_cuda
    assert float((grouped - loop).abs().max()) < 5e-5 and float((small - loop).abs().max()) < 5e-5
    assert torch.equal(extract_vectors(net, imgs[::-1], cuda_device), grouped.flip(1))        # same variant class: bitwise, order-independent
    assert torch.equal(extract_vectors(net, imgs, cuda_device), grouped)                        # deterministic
    assert torch.allclose(grouped.norm(dim=0), torch.ones(64, device=cuda_device), atol=1e-5)
    assert float(torch.nn.functional.cosine_similarity(grouped.t(), loop.t(), dim=1).min()) > 0.999999


def test_extract_vectors_many_distinct_sizes_runs_groups_concurrently(cuda_device, tmp_path):
    """a collection whose sizes hardly repeat (longer side fixed, arbitrary aspect): the groups are one or two images each, and extract_vectors hands them to the
    network together, eight images in flight (SingleNetwork.forward_list: every pyramid level of every group in ONE forward_many).  Same columns as with concurrent=1 bit for bit
    (same kernels per forward, own scratch buffer per forward), and as the image-by-image loop to 5e-5; forward_list == [net(x) for x] bitwise."""
    import pickle
    import hubconf
    from gandtr_amd.learning.checkpoints import Checkpoints
    from gandtr_amd.stages.validate import extract_vectors
    import gandtr_amd.learning.network as NW
    base = hubconf.gem_vgg16_cyclegan(pretrained=False, device="cpu")
    base.model.load_state_dict(synth.vgg16_state(0))
    sd = base.state_dict()["net"]
    sd["network_params"]["runtime"]["data"] = {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]}
    ck, lw = str(tmp_path / "vgg.pth"), str(tmp_path / "lw.pkl")
    torch.save(sd, ck)
    with open(lw, "wb") as f:
        pickle.dump(synth.whitening_state(0, 512), f)
    runtime = {"wrappers": {"train": None, "eval": {"0_cirwhiten": {"whitening": lw, "dimensions": None}, "1_cirmultiscale": {"scales": True}}}}
    net = NW.initialize_network(None, cuda_device, Checkpoints.load_network(ck), runtime).eval()
    shorts = [80, 88, 96, 104, 112, 120, 128]
    sizes = [((128, shorts[i % 7]) if i % 3 else (shorts[(2 * i) % 7], 128)) for i in range(22)]
    imgs = [synth.synth_input(300 + i, (3,) + sz) for i, sz in enumerate(sizes)]
    assert len(set(sizes)) >= 12
    one = extract_vectors(net, imgs, cuda_device, concurrent=1)
    four = extract_vectors(net, imgs, cuda_device)                       # default: small groups together, eight images in flight
    loop = extract_vectors(net, imgs, cuda_device, batched=False)
    assert torch.equal(one, four) and four.shape == (512, 22)
    assert float((four - loop).abs().max()) < 5e-5
    xs = [im.unsqueeze(0) for im in imgs[:5]]
    with torch.no_grad():
        listed = net.forward_list(xs)
        for x, y in zip(xs, listed):
            assert torch.equal(net(x), y)



def test_fused_statistics_with_an_odd_number_of_128_row_records(cuda_device):
    """InstanceNorm statistics from a conv epilogue whose 256-row tiles do not divide the layer (N*OH*OW = 128 * odd): the last tile
    owns ONE record, not two (the slab holds M / 128 records; conv_epilogue.h).  The normalised output must be right and the tensor
    allocated right behind the slab must stay intact."""
    net = HipNet(cuda_device, "f16")
    t = net.input(3)
    w = _g(0, "w", (64, 3, 7, 7), 0.1)
    raw = net.conv(t, w, None, pad=3, reflect=True)
    out = net.instance_norm(raw, relu=True)
    slot = net.output_nchw(out)
    net.finalize()
    x = synth.synth_input(5, (1, 3, 384, 683), 1.0)               # M = 262272 = 128 * 2049; width not a multiple of 32: generic kernel
    got = net.forward(x.to(cuda_device))[slot].cpu()
    ref = F.relu(F.instance_norm(F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), w.half().float()), eps=1e-5))
    assert got.shape == ref.shape and _rel(got, ref) < 4e-3


def test_config5_full_size_chain(cuda_device):
    """BASELINE config 5 at its full per-rank size: cyclegan on 128 x 3 x 256 x 256 -> meanstd_post -> GeM-ResNet-101 descriptors
    through the container API; two of the 128 images against the CPU oracle (north_star gates), all columns unit-norm"""
    from gandtr_amd.learning import network as N
    from oracle import gandtr_oracle as O
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
    chain = N.initialize_network({"type": "CirSequentialNetwork", "sequence": "augment,embed", "augment": gen, "embed": emb}, cuda_device).eval()
    gsd, esd = synth.generator_state(0, "instance", gain=0.02), synth.resnet101_state(0, p=3.0)
    chain.networks["augment"].model.load_state_dict(gsd)
    chain.networks["embed"].model.load_state_dict(esd)
    x = synth.synth_input(9, (128, 3, 256, 256), 1.0)
    with torch.no_grad():
        d = chain(x.to(cuda_device))
    assert d.shape == (2048, 128)
    assert torch.allclose(d.norm(dim=0), torch.ones(128, device=d.device), atol=1e-5)
    for i in (3, 127):
        y = O.resnet_generator(x[i:i + 1], gsd, "instance", 9)
        y = (y * 0.5 + 0.5 - torch.tensor([0.485, 0.456, 0.406])[:, None, None]) / torch.tensor([0.229, 0.224, 0.225])[:, None, None]
        ref = O.image_retrieval_forward(y, esd, "resnet101")[:, 0]
        got = d[:, i].cpu()
        assert float(torch.nn.functional.cosine_similarity(got, ref, dim=0)) >= 0.9999
        assert float((got - ref).abs().max()) <= 1e-3


def test_extract_vectors_from_files_equals_the_pillow_pipeline(cuda_device, tmp_path):
    """the reference's extract_vectors takes image PATHS (imageretrievalnet.py:312-339: ImagesFromList -> pil_loader -> imresize ->
    transform, batch 1).  extract_vectors_from_files does decode + resize + normalise on the device; against tensors prepared the
    reference's way (Pillow decode, thumbnail, numpy normalise) and the same network, every descriptor agrees to 5e-5 -- the decoded
    and resized pixels are identical (tests/test_hip_jpeg.py, test_hip_ingest.py), what is left is the float normalisation's rounding."""
    import io
    import numpy as np
    from PIL import Image
    from gandtr_amd.stages.validate import extract_vectors, extract_vectors_from_files
    import copy
    import gandtr_amd.learning as L
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    emb = {"type": "SingleNetwork",
           "model": {"architecture": "cirnet", "cir_architecture": "vgg16", "local_whitening": False, "pooling": "gem",
                     "pretrained": False, "regional": False, "whitening": False},
           "initialize": False, "path": None,
           "runtime": {"wrappers": "cirfaketuplebatch", "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [mean, std]}}}
    cpu_net = L.load_network(copy.deepcopy(emb), "cpu")
    cpu_net.model.load_state_dict(synth.vgg16_state(0))
    torch.save(cpu_net.state_dict()["net"], tmp_path / "vgg.pth")
    net = L.load_network({"path": str(tmp_path / "vgg.pth"), "runtime": copy.deepcopy(emb["runtime"])}, cuda_device).eval()
    rng = np.random.RandomState(0)
    blobs, want = [], []
    for i, (w, h) in enumerate([(300, 200), (200, 300), (300, 200), (260, 260), (200, 300), (300, 200), (90, 120)]):
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([128 + 100 * np.sin(xx / (9.0 + c + i)) * np.cos(yy / (7.0 + i)) + rng.normal(0, 12, (h, w)) for c in range(3)], -1)
        buf = io.BytesIO()
        Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=88, subsampling=(2, 0, 1)[i % 3])
        blobs.append(buf.getvalue())
        img = Image.open(io.BytesIO(blobs[-1])).convert("RGB")
        img.thumbnail((160, 160), Image.LANCZOS)
        a = (np.asarray(img).astype(np.float32) / 255.0 - np.array(mean, np.float32)) / np.array(std, np.float32)
        want.append(torch.from_numpy(a.transpose(2, 0, 1).copy()))
    got = extract_vectors_from_files(net, blobs, 160, (mean, std), cuda_device, chunk=4)
    ref = extract_vectors(net, want, cuda_device)
    assert got.shape == ref.shape == (512, 7)
    assert float((got - ref).abs().max()) < 5e-5 and float(torch.nn.functional.cosine_similarity(got.t(), ref.t(), dim=1).min()) > 0.999999


def test_rank_images_on_jpeg_files(cuda_device, tmp_path):
    """the retrieval stage on image FILES, as the reference's datasets hold them: paths and in-memory file contents go through the device
    decoder + resize + normalise (extract_vectors_from_files); a query that is a re-encoded copy of a database image ranks it first"""
    import copy
    import io
    from PIL import Image
    from gandtr_amd.stages.validate import rank_images
    import gandtr_amd.learning as L
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    emb = {"type": "SingleNetwork",
           "model": {"architecture": "cirnet", "cir_architecture": "vgg16", "local_whitening": False, "pooling": "gem",
                     "pretrained": False, "regional": False, "whitening": False},
           "initialize": False, "path": None,
           "runtime": {"wrappers": "cirfaketuplebatch", "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [mean, std]}}}
    net = L.load_network(copy.deepcopy(emb), "cpu")
    net.model.load_state_dict(synth.vgg16_state(0))
    torch.save(net.state_dict()["net"], tmp_path / "vgg.pth")
    params = {"network": {"path": str(tmp_path / "vgg.pth"), "runtime": copy.deepcopy(emb["runtime"])}}
    rng = np.random.RandomState(1)
    paths, pictures = [], []
    for i in range(6):
        h, w = (180, 240) if i % 2 else (240, 180)
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([128 + 90 * np.sin(xx / (6.0 + 2 * i + c)) * np.cos(yy / (5.0 + i)) + rng.normal(0, 10, (h, w)) for c in range(3)], -1)
        pictures.append(Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8)))
        paths.append(str(tmp_path / ("db%d.jpg" % i)))
        pictures[-1].save(paths[-1], "JPEG", quality=90, subsampling=i % 3)
    buf = io.BytesIO()
    pictures[4].save(buf, "JPEG", quality=60)                       # the query: image 4 again, other quality (bytes, not a path)
    meta, ranks, scores = rank_images(params, (paths, [buf.getvalue()], {"image_size": 160}))
    assert ranks.shape == (6, 1) and ranks[0, 0] == 4 and scores[4, 0] > 0.97
    assert meta["eval"]["database"] == 6 and meta["eval"]["queries"] == 1


def test_infer_stage_groups_equal_sizes_on_the_device(cuda_device, tmp_path, monkeypatch):
    """the `infer` stage on a HIP device: items of equal size go through the generator as one batch (mdir/stages/infer.py:17-66 runs them one by one; no op of the
    generator crosses images) -- same pictures as the item-by-item loop (GANDTR_INFER_BATCH=1) to the f16c mode's own tolerance, in the input's order"""
    import hubconf
    from gandtr_amd.stages import FUNCTIONS
    infer = FUNCTIONS["mdir.stages.infer.infer"]
    gen = hubconf.cyclegan(pretrained=False, device="cpu")
    gen.model.load_state_dict(synth.generator_state(0, "instance"))
    sd = gen.state_dict()["net"]
    sd["network_params"]["runtime"]["data"] = {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}
    ck = tmp_path / "gen.pth"
    torch.save(sd, ck)
    params = {"network": {"path": str(ck), "runtime": {"wrappers": ""}}, "output": {"inference": {"name": "rgb"}}}
    items = [synth.synth_input(700 + i, (3, 64, 64) if i % 3 else (3, 64, 96), 1.0) for i in range(11)]
    meta, grouped = infer(params, (items,))
    monkeypatch.setenv("GANDTR_INFER_BATCH", "1")
    _, loop = infer(params, (items,))
    assert meta["stats"]["items"] == 11 and len(grouped) == len(loop) == 11
    for g, l, x in zip(grouped, loop, items):
        assert g.shape == l.shape == (x.shape[1], x.shape[2], 3)
        assert float(np.abs(g - l).max()) < 2e-3          # pictures in [0, 1]; batch 1 and batch 4-7 pick other kernel variants (f16x3 generic vs compensated patch kernels)
    # a batch is bounded by its pixels, too (here: two 64 x 64 items, one 64 x 96 item per forward) -- same pictures, input order kept
    monkeypatch.setenv("GANDTR_INFER_BATCH", "64")
    monkeypatch.setenv("GANDTR_INFER_PIXELS", str(2 * 64 * 64))
    _, capped = infer(params, (items,))
    for c, l in zip(capped, loop):
        assert c.shape == l.shape and float(np.abs(c - l).max()) < 2e-3
