"""GPU parity of single layers / small graphs built through the C ABI against torch fp32 CPU ops (the oracle's
building blocks).  Tolerances: the HIP path stores activations in fp16 with fp32 accumulation, so a single layer is
held to 2e-3 of the reference's max magnitude (fp16 rounding of inputs, weights and outputs: 3 x 2^-11)."""
import math

import pytest
import torch
import torch.nn.functional as F

from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _run_single_conv(dev, cin, cout, k, stride, pad, reflect, transposed, relu, bn, n=2, h=20, w=28, seed=0):
    from gandtr_amd.engine import HipNet
    g = lambda name, shape, std=1.0: synth._normal(seed, name, shape, std)
    net = HipNet(dev)
    t = net.input(3)
    # lift 3 -> cin channels with a 1x1 conv so that the layer under test sees a real multi-channel fp16 input
    w0 = g("w0", (cin, 3, 1, 1), 0.7)
    t = net.conv(t, w0)
    tap_in = net.output_nchw(t)
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = g("w", wshape, math.sqrt(2.0 / (cin * k * k)))
    bias = g("b", (cout,), 0.2)
    bnp = None
    if bn:
        bnp = (synth._uniform(seed, "g", (cout,), 0.5, 1.5), g("be", (cout,), 0.2), g("m", (cout,), 0.2),
               synth._uniform(seed, "v", (cout,), 0.5, 1.5))
    out = net.conv(t, wt, bias, bn=bnp, stride=stride, pad=pad, reflect=reflect, transposed=transposed, relu=relu)
    tap_out = net.output_nchw(out)
    net.finalize()
    x = synth.synth_input(seed + 1, (n, 3, h, w))
    outs = net.forward(x.to(dev))
    xin = outs[tap_in].cpu()      # the fp16-rounded activations the layer really consumed
    wq = wt.half().float()
    if transposed:
        ref = F.conv_transpose2d(xin, wq, bias, stride=2, padding=1, output_padding=1)
    else:
        xi = F.pad(xin, (pad,) * 4, mode="reflect") if reflect else xin
        ref = F.conv2d(xi, wq, bias, stride=stride, padding=0 if reflect else pad)
    if bn:
        # BN is folded into the fp16 weights on the device, so fold it the same way for a tight comparison
        ref = F.batch_norm(ref, bnp[2], bnp[3], bnp[0], bnp[1], training=False, eps=1e-5)
    if relu:
        ref = F.relu(ref)
    return outs[tap_out].cpu(), ref


@pytest.mark.parametrize("cfg", [
    # cin, cout, k, stride, pad, reflect, transposed, relu, bn
    (8, 8, 3, 1, 1, True, False, False, False),
    (16, 32, 3, 1, 1, False, False, True, False),
    (64, 64, 3, 1, 1, True, False, True, True),
    (64, 128, 3, 2, 1, False, False, True, False),
    (128, 256, 3, 2, 1, False, False, False, True),
    (256, 256, 3, 1, 1, True, False, False, False),
    (256, 128, 3, 2, 1, False, True, True, False),
    (32, 16, 3, 2, 1, False, True, False, True),
    (64, 256, 1, 1, 0, False, False, True, True),
    (256, 64, 1, 2, 0, False, False, False, True),
    (8, 64, 7, 1, 3, True, False, True, False),
    (8, 64, 7, 2, 3, False, False, True, True),
    (512, 512, 3, 1, 1, False, False, True, False),
])
def test_conv_layer(cuda_device, cfg):
    cin, cout, k, stride, pad, reflect, transposed, relu, bn = cfg
    got, ref = _run_single_conv(cuda_device, cin, cout, k, stride, pad, reflect, transposed, relu, bn)
    assert got.shape == ref.shape
    tol = 2e-3 if not bn else 4e-3     # folded BN scales the fp16 weight rounding
    assert _rel(got, ref) < tol, (cfg, _rel(got, ref))


@pytest.mark.parametrize("cfg", [
    # cin, cout, reflect, relu, bn, (n, h, w)          -- sizes with >= 512 (patch, N tile) pairs: conv3x3_halo_rb.hip runs
    (256, 256, False, True, False, (8, 128, 128)),     # VGG-style zero padding + bias + ReLU
    (256, 256, True, False, False, (9, 60, 128)),      # ragged patch rows (60 = 3 x 16 + 12), reflect padding
    (128, 512, False, True, True, (8, 96, 96)),        # two N tiles, folded BN, 2 input chunks
    (64, 256, False, False, False, (32, 64, 64)),      # single 64-channel chunk (no chunk barrier between tiles)
])
def test_conv3x3_persistent_halo_kernel(cuda_device, cfg):
    cin, cout, reflect, relu, bn, (n, h, w) = cfg
    got, ref = _run_single_conv(cuda_device, cin, cout, 3, 1, 1, reflect, False, relu, bn, n=n, h=h, w=w)
    assert got.shape == ref.shape
    assert _rel(got, ref) < (4e-3 if bn else 2e-3), (cfg, _rel(got, ref))


@pytest.mark.parametrize("cin,cout,shape", [(256, 256, (8, 128, 128)), (64, 64, (8, 128, 128)), (128, 128, (9, 60, 128)), (256, 512, (8, 96, 96)),
                                            (64, 64, (4, 150, 171)), (256, 256, (8, 127, 128)), (128, 128, (4, 109, 121))])      # odd sizes: the pool drops the last row / column
def test_conv_relu_maxpool_fused(cuda_device, cin, cout, shape):
    """VGG16 stage ends (Conv2d 3x3 + ReLU + MaxPool2d(2, 2), torchvision cfg "D"): the pool runs in the conv epilogue of the
    patch kernels (conv3x3_halo_rb.hip / conv_epilogue.h) and the full-resolution tensor is never written."""
    from gandtr_amd.engine import HipNet
    n, h, w = shape
    net = HipNet(cuda_device)
    t = net.input(3)
    a = net.conv(t, synth._normal(0, "w0", (cin, 3, 1, 1), 0.7))
    tap_in = net.output_nchw(a)
    wt, b = synth._normal(0, "w", (cout, cin, 3, 3), math.sqrt(2.0 / (cin * 9))), synth._normal(0, "b", (cout,), 0.2)
    o = net.maxpool(net.conv(a, wt, b, pad=1, relu=True), 2, 2)
    tap = net.output_nchw(o)
    net.finalize()
    x = synth.synth_input(7, (n, 3, h, w))
    outs = net.forward(x.to(cuda_device))
    ref = F.max_pool2d(F.relu(F.conv2d(outs[tap_in].cpu(), wt.half().float(), b, padding=1)), 2, 2)
    got = outs[tap].cpu()
    assert got.shape == ref.shape
    assert _rel(got, ref) < 2e-3


def test_conv_ragged_tail(cuda_device):
    """M not a multiple of the 128-row tile and odd spatial sizes."""
    got, ref = _run_single_conv(cuda_device, 64, 64, 3, 1, 1, True, False, True, False, n=3, h=13, w=17)
    assert _rel(got, ref) < 2e-3
    got, ref = _run_single_conv(cuda_device, 64, 128, 3, 2, 1, False, False, False, False, n=1, h=9, w=11)
    assert _rel(got, ref) < 2e-3


@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 37, 53), (1, 8, 32)])
def test_generator_head_7x7_tanh(cuda_device, norm, shape):
    """ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh (p2p_networks.py:433-436), fp32 NCHW output: the fused head kernel
    (conv_head7.hip), alone and with the preceding InstanceNorm + ReLU folded into its input staging; ragged tiles."""
    from gandtr_amd.engine import HipNet
    n, h, w = shape
    net = HipNet(cuda_device)
    t = net.input(3)
    t = net.conv(t, synth._normal(0, "w0", (64, 3, 3, 3), 0.4), pad=1, reflect=True)
    tap_in = net.output_nchw(t)
    u = net.instance_norm(t, relu=True) if norm else t
    wt, b = synth._normal(0, "wh", (3, 64, 7, 7), 0.03), synth._normal(0, "bh", (3,), 0.2)
    o = net.conv(u, wt, b, pad=3, reflect=True, act=1, out_f32=True)          # act 1 = tanh; returns the output slot
    net.finalize()
    x = synth.synth_input(2, (n, 3, h, w))
    outs = net.forward(x.to(cuda_device))
    xin = outs[tap_in].cpu()
    if norm:
        xin = F.relu(F.instance_norm(xin, eps=1e-5)).half().float()      # the head consumes fp16-rounded activations
    ref = torch.tanh(F.conv2d(F.pad(xin, (3,) * 4, mode="reflect"), wt.half().float(), b))
    got = outs[o].cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < (3e-3 if norm else 1.5e-3)


@pytest.mark.parametrize("cfg", [
    # k, stride, reflect, relu, bn, norm_after, (n, h, w)
    (7, 1, True, False, False, True, (4, 128, 128)),      # generator stem + InstanceNorm statistics from its epilogue
    (7, 1, True, False, False, False, (3, 150, 171)),     # ragged tiles
    (7, 2, False, True, True, False, (2, 384, 400)),      # ResNet-101 stem (BN folded, ReLU)
    (3, 1, False, True, False, False, (2, 200, 180)),     # VGG16 conv1_1
])
def test_image_stem_kernels(cuda_device, cfg):
    """conv_stem.hip: image (3 -> 8 channels) to 64 channels, 7x7 / 3x3, against torch on the fp16-rounded image."""
    from gandtr_amd.engine import HipNet
    k, stride, reflect, relu, bn, norm_after, (n, h, w) = cfg
    net = HipNet(cuda_device)
    t = net.input(3)
    wt = synth._normal(0, "ws", (64, 3, k, k), math.sqrt(2.0 / (3 * k * k)))
    bias = synth._normal(0, "bs", (64,), 0.2)
    bnp = None
    if bn:
        bnp = (synth._uniform(0, "g", (64,), 0.5, 1.5), synth._normal(0, "be", (64,), 0.2), synth._normal(0, "m", (64,), 0.2),
               synth._uniform(0, "v", (64,), 0.5, 1.5))
    o = net.conv(t, wt, bias, bn=bnp, stride=stride, pad=k // 2, reflect=reflect, relu=relu)
    tap = net.output_nchw(o)
    tap_n = net.output_nchw(net.instance_norm(o, relu=True)) if norm_after else None
    net.finalize()
    x = synth.synth_input(4, (n, 3, h, w))
    outs = net.forward(x.to(cuda_device))
    xin = x.half().float()
    xi = F.pad(xin, (k // 2,) * 4, mode="reflect") if reflect else xin
    ref = F.conv2d(xi, wt.half().float(), bias, stride=stride, padding=0 if reflect else k // 2)
    if bn:
        ref = F.batch_norm(ref, bnp[2], bnp[3], bnp[0], bnp[1], training=False, eps=1e-5)
    if relu:
        ref = F.relu(ref)
    got = outs[tap].cpu()
    assert got.shape == ref.shape
    assert _rel(got, ref) < (4e-3 if bn else 2e-3)
    if norm_after:
        assert _rel(outs[tap_n].cpu(), F.relu(F.instance_norm(got, eps=1e-5))) < 2e-3


@pytest.mark.parametrize("cin,cout,fold", [(128, 64, True), (256, 128, False), (128, 64, False), (256, 128, True), (256, 128, "res")])
def test_transposed_conv_fused_phases(cuda_device, cin, cout, fold):
    """ConvTranspose2d(k3, s2, p1, op1) (p2p_networks.py:289-300) as ONE GEMM over the four sub-pixel phases
    (conv_igemm_rb.hip, phase_cout): plain, with the InstanceNorm statistics of its output taken in the epilogue, and with
    the producer's InstanceNorm + ReLU folded into its input staging."""
    from gandtr_amd.engine import HipNet
    net = HipNet(cuda_device)
    t = net.input(3)
    a = net.conv(t, synth._normal(0, "w0", (cin, 3, 3, 3), 0.4), pad=1, reflect=True)
    tap_in = net.output_nchw(a)
    if fold == "res":          # y = r + IN(a) feeding only the transposed conv (last ResnetBlock -> first up-sampling conv)
        r = net.conv(t, synth._normal(0, "w1", (cin, 3, 3, 3), 0.4), pad=1, reflect=True)
        tap_r = net.output_nchw(r)
        u = net.instance_norm(a, relu=False, residual=r)
    else:
        u = net.instance_norm(a, relu=True) if fold else a
    wt, b = synth._normal(0, "wt", (cin, cout, 3, 3), math.sqrt(2.0 / (cin * 2.25))), synth._normal(0, "bt", (cout,), 0.2)
    o = net.conv(u, wt, b, stride=2, pad=1, transposed=True)
    tap = net.output_nchw(o)
    tap_n = net.output_nchw(net.instance_norm(o, relu=True))
    net.finalize()
    x = synth.synth_input(6, (32, 3, 64, 64))          # 512 tiles of 256 GEMM rows: the persistent kernel is eligible
    outs = net.forward(x.to(cuda_device))
    xin = outs[tap_in].cpu()
    if fold == "res":
        xin = (outs[tap_r].cpu() + F.instance_norm(xin, eps=1e-5)).half().float()
    elif fold:
        xin = F.relu(F.instance_norm(xin, eps=1e-5)).half().float()
    ref = F.conv_transpose2d(xin, wt.half().float(), b, stride=2, padding=1, output_padding=1)
    got = outs[tap].cpu()
    assert got.shape == ref.shape
    assert _rel(got, ref) < (3e-3 if fold else 2e-3)
    assert _rel(outs[tap_n].cpu(), F.relu(F.instance_norm(got, eps=1e-5))) < 2e-3


def test_resnet_block_chain_folds(cuda_device):
    """Three ResnetBlocks (p2p_networks.py:480-505) at 256 channels behind an InstanceNorm whose output feeds both the first
    conv and the first residual: exercises the folded norm with write-back (norm only, and norm + residual) of
    conv3x3_halo_rb.hip and the plain apply pass for the final block output."""
    from gandtr_amd.engine import HipNet
    net = HipNet(cuda_device)
    t = net.input(3)
    raw = net.conv(t, synth._normal(0, "w0", (256, 3, 3, 3), 0.3), pad=1, reflect=True)
    tap = net.output_nchw(raw)
    y = net.instance_norm(raw, relu=True)
    ws = []
    for i in range(3):
        w1, w2 = synth._normal(0, "a%d" % i, (256, 256, 3, 3), 0.03), synth._normal(0, "b%d" % i, (256, 256, 3, 3), 0.03)
        ws.append((w1, w2))
        u = net.instance_norm(net.conv(y, w1, pad=1, reflect=True), relu=True)
        y = net.instance_norm(net.conv(u, w2, pad=1, reflect=True), relu=False, residual=y)
    out = net.output_nchw(y)
    net.finalize()
    x = synth.synth_input(5, (32, 3, 64, 64))          # 32 images x 16 patches = 512 tiles: the persistent halo kernel is eligible
    outs = net.forward(x.to(cuda_device))
    r = F.relu(F.instance_norm(outs[tap].cpu(), eps=1e-5))
    q = lambda v: v.half().float()
    for w1, w2 in ws:
        u = F.relu(F.instance_norm(F.conv2d(F.pad(q(r), (1,) * 4, mode="reflect"), q(w1)), eps=1e-5))
        r = r + F.instance_norm(F.conv2d(F.pad(q(u), (1,) * 4, mode="reflect"), q(w2)), eps=1e-5)
    assert _rel(outs[out].cpu(), r) < 4e-3


def test_instance_norm_relu_residual(cuda_device):
    from gandtr_amd.engine import HipNet
    net = HipNet(cuda_device)
    t = net.input(3)
    a = net.conv(t, synth._normal(0, "wa", (32, 3, 1, 1), 0.7))
    b = net.conv(t, synth._normal(0, "wb", (32, 3, 1, 1), 0.7))
    ta, tb = net.output_nchw(a), net.output_nchw(b)
    o1 = net.output_nchw(net.instance_norm(a, relu=True))
    o2 = net.output_nchw(net.instance_norm(a, relu=False, residual=b))
    net.finalize()
    x = synth.synth_input(3, (2, 3, 40, 24))
    outs = [o.cpu() for o in net.forward(x.to(cuda_device))]
    A, B = outs[ta], outs[tb]
    assert _rel(outs[o1], F.relu(F.instance_norm(A, eps=1e-5))) < 1.5e-3
    assert _rel(outs[o2], F.instance_norm(A, eps=1e-5) + B) < 1.5e-3


@pytest.mark.parametrize("k,s,p", [(2, 2, 0), (3, 2, 1)])
def test_maxpool(cuda_device, k, s, p):
    from gandtr_amd.engine import HipNet
    net = HipNet(cuda_device)
    t = net.input(3)
    a = net.conv(t, synth._normal(0, "wa", (16, 3, 1, 1), 0.7))
    ta = net.output_nchw(a)
    to = net.output_nchw(net.maxpool(a, k, s, p))
    net.finalize()
    x = synth.synth_input(4, (2, 3, 21, 30))
    outs = [o.cpu() for o in net.forward(x.to(cuda_device))]
    assert torch.equal(outs[to], F.max_pool2d(outs[ta], k, s, p))    # max of fp16 values is exact


@pytest.mark.parametrize("scale", [None, 1.0, 1.0 / math.sqrt(2), 0.5, math.sqrt(2)])
def test_input_pack_and_resize(cuda_device, scale):
    """F.interpolate(scale_factor=s, bilinear, align_corners=False) fused into the input pack kernel (wrapper.py:225)."""
    from gandtr_amd.engine import HipNet
    net = HipNet(cuda_device)
    t = net.input(3)
    o = net.output_nchw(t)
    net.finalize()
    x = synth.synth_input(5, (2, 3, 40, 56))
    got = net.forward(x.to(cuda_device), scale=scale)[o].cpu()[:, :3]
    ref = x if scale is None else F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 4e-3      # fp16 storage of values up to ~4


def test_input_perm_affine(cuda_device):
    from gandtr_amd.engine import HipNet
    net = HipNet(cuda_device)
    t = net.input(3, perm=[2, 1, 0], scale=[0.5, 2.0, 1.0], shift=[0.1, -0.2, 0.3])
    o = net.output_nchw(t)
    net.finalize()
    x = synth.synth_input(6, (1, 3, 8, 8))
    got = net.forward(x.to(cuda_device))[o].cpu()[:, :3]
    ref = x[:, [2, 1, 0]] * torch.tensor([0.5, 2.0, 1.0])[None, :, None, None] + torch.tensor([0.1, -0.2, 0.3])[None, :, None, None]
    assert float((got - ref).abs().max()) < 4e-3


@pytest.mark.parametrize("pool,n,h,w", [(True, 1, 530, 614), (True, 3, 224, 448), (False, 1, 530, 614), (True, 2, 512, 512)])
def test_resnet_stem_from_the_fp32_image(cuda_device, pool, n, h, w):
    """conv_stem_pair[_pool]_kernel: Conv2d(3, 64, 7, stride 2, pad 3) + BN + ReLU (+ MaxPool2d(3, 2, 1)) read straight from the caller's fp32
    NCHW image with the input op's channel permutation / scale / shift applied in the loader (no pack launch, no packed tensor), ragged tile
    edges in both directions (pooled 133 x 154 is not a multiple of the 7 x 15 tile), against fp64 on the fp16-rounded transformed image; and
    that these are the kernels that ran.  A call that resizes (multi-scale pyramid) must take the general path and agree with torch as well."""
    from gandtr_amd.engine import HipNet
    perm, scale, shift = [2, 1, 0], [1.7, 0.6, -1.2], [0.3, -0.2, 0.1]
    net = HipNet(cuda_device)
    t = net.input(3, perm=perm, scale=scale, shift=shift)
    wt = synth._normal(0, "ws", (64, 3, 7, 7), math.sqrt(2.0 / 147))
    bnp = (synth._uniform(0, "g", (64,), 0.5, 1.5), synth._normal(0, "be", (64,), 0.2), synth._normal(0, "m", (64,), 0.2), synth._uniform(0, "v", (64,), 0.5, 1.5))
    o = net.conv(t, wt, None, bn=bnp, stride=2, pad=3, relu=True)
    if pool:
        o = net.maxpool(o, 3, 2, 1)
    tap = net.output_nchw(o)
    net.finalize()
    x = synth.synth_input(9, (n, 3, h, w))
    net.set_profiling(True)
    got = net.forward(x.to(cuda_device))[tap].cpu()
    torch.cuda.synchronize()
    variants = [v for k, v, ms, fl in net.profile() if k == 1]
    assert variants[0] == (952049 if pool else 951049), variants

    def reference(img):
        xin = (img[:, perm] * torch.tensor(scale).view(1, 3, 1, 1) + torch.tensor(shift).view(1, 3, 1, 1)).half().double()
        r = F.conv2d(xin, wt.half().double(), None, stride=2, padding=3)        # (the packer folds BN before rounding the weights: compare at 4e-3 as the stem test above)
        r = F.relu(F.batch_norm(r, bnp[2].double(), bnp[3].double(), bnp[0].double(), bnp[1].double(), training=False, eps=1e-5))
        return F.max_pool2d(r, 3, 2, 1) if pool else r

    ref = reference(x)
    assert got.shape == ref.shape
    assert _rel(got.double(), ref) < 4e-3
    # borders: first / last pooled rows and columns see the conv's zero padding and the pool's implicit one
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), slice(None), slice(None), 0), (slice(None), slice(None), slice(None), -1)):
        assert float((got.double()[sl] - ref[sl]).abs().max() / ref.abs().max()) < 4e-3
    assert torch.equal(got, net.forward(x.to(cuda_device))[tap].cpu())
    # resized call (bilinear 1/sqrt2 inside the pack kernel): general path, same network
    net.set_profiling(True)
    rs = net.forward(x.to(cuda_device), scale=2 ** -0.5)[tap].cpu()
    torch.cuda.synchronize()
    assert [v for k, v, ms, fl in net.profile() if k == 1][0] not in (951049, 952049)
    small = F.interpolate(x, scale_factor=2 ** -0.5, mode="bilinear", align_corners=False, recompute_scale_factor=False)
    ref_s = reference(small)
    assert rs.shape == ref_s.shape and _rel(rs.double(), ref_s) < 4e-3


@pytest.mark.parametrize("n,h,w", [(1, 270, 310), (2, 256, 512)])
def test_vgg_first_conv_from_the_fp32_image(cuda_device, n, h, w):
    """conv_stem_pair_kernel<3, 1>: Conv2d(3, 64, 3, pad 1) + ReLU (VGG16 conv1_1, the HED trunk's first conv) read straight from the caller's
    fp32 NCHW image with the wrappers' BGR permutation and mean / std folded in; ragged 16 x 32 tiles; against fp64."""
    from gandtr_amd.engine import HipNet
    perm, scale, shift = [2, 1, 0], [255.0, 255.0, 255.0], [-104.0, -117.0, -123.0]
    net = HipNet(cuda_device)
    t = net.input(3, perm=perm, scale=scale, shift=shift)
    wt, b = synth._normal(0, "wv", (64, 3, 3, 3), 0.004), synth._normal(0, "bv", (64,), 0.2)
    tap = net.output_nchw(net.conv(t, wt, b, pad=1, relu=True))
    net.finalize()
    x = synth.synth_input(11, (n, 3, h, w)).abs().clamp(0, 1)
    net.set_profiling(True)
    got = net.forward(x.to(cuda_device))[tap].cpu()
    torch.cuda.synchronize()
    assert [v for k, v, ms, fl in net.profile() if k == 1][0] == 951009
    xin = (x[:, perm] * torch.tensor(scale).view(1, 3, 1, 1) + torch.tensor(shift).view(1, 3, 1, 1)).half().double()
    ref = F.relu(F.conv2d(xin, wt.half().double(), b.double(), padding=1))
    assert got.shape == ref.shape and _rel(got.double(), ref) < 2e-3
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), -1), (slice(None), slice(None), slice(None), 0), (slice(None), slice(None), slice(None), -1)):
        assert float((got.double()[sl] - ref[sl]).abs().max() / ref.abs().max()) < 2e-3


@pytest.mark.parametrize("cin,cout,reflect,pool,shape", [
    (64, 64, False, True, (2, 150, 176)),       # tall 16 x 32 patches, last patch row 22 of 32 deep, single-stage form, fused pool
    (64, 64, True, False, (4, 112, 90)),        # reflect padding, ragged in both directions (112 = 3.5 patches of 32, 90 = 5.6 of 16)
    (64, 128, False, False, (2, 150, 170)),     # single-stage form, 2 x 2 waves
    (128, 128, False, True, (4, 108, 120)),     # two chunks: double-buffered four-wave form, fused pool
    (128, 128, True, False, (2, 150, 170)),
])
def test_conv3x3_small_channel_forms_of_the_patch_kernel(cuda_device, cin, cout, reflect, pool, shape):
    """conv3x3_halo_rb.hip's four-wave forms for 64 / 128 output channels (VGG16 conv1_2 / conv2_x, the HED trunk): ragged patch edges, reflect
    padding, the fused 2 x 2 max pool, against fp64 on the fp16 tensors the layer consumed; and that these forms are the ones that ran."""
    from gandtr_amd.engine import HipNet
    n, h, w = shape
    net = HipNet(cuda_device)
    t = net.input(3)
    a = net.conv(t, synth._normal(0, "w0", (cin, 3, 1, 1), 0.7), relu=True)
    tap_in = net.output_nchw(a)
    wt, b = synth._normal(0, "w", (cout, cin, 3, 3), math.sqrt(2.0 / (cin * 9))), synth._normal(0, "b", (cout,), 0.2)
    o = net.conv(a, wt, b, pad=1, reflect=reflect, relu=True)
    if pool:
        o = net.maxpool(o, 2, 2)
    tap = net.output_nchw(o)
    net.finalize()
    x = synth.synth_input(8, (n, 3, h, w))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    assert 910000 + cout in [v for k, v, ms, fl in net.profile() if k == 1]
    xin = outs[tap_in].double().cpu()
    xi = F.pad(xin, (1,) * 4, mode="reflect") if reflect else xin
    ref = F.relu(F.conv2d(xi, wt.half().double(), b.double(), padding=0 if reflect else 1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    got = outs[tap].double().cpu()
    assert got.shape == ref.shape and _rel(got, ref) < 2e-3
    assert torch.equal(outs[tap], net.forward(x.to(cuda_device))[tap])
