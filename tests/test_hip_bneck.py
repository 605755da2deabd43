"""GPU parity of the fused identity Bottleneck (conv_bneck.hip: conv 1x1 -> 3x3 -> 1x1 + residual, torchvision's Bottleneck as restated in
oracle/gandtr_oracle.py:116-133, one launch; ResNet-101 layer1 / layer2 shapes) against an fp64 evaluation of the same block on the same
fp16-rounded input with the two intermediate tensors rounded to fp16 where the kernel stores them; against the layer-by-layer kernels of
the same build; image borders (the 3x3 conv zero-pads r, not x); and that the fused kernel is the one that ran."""
import pytest
import torch
import torch.nn.functional as F

from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu


def _g(name, shape, std):
    return synth._normal(0, name, shape, std)


def _block_net(dev, C, mid, nblocks=1):
    net = HipNet(dev, "f16")
    t = net.input(3)
    x = net.conv(t, _g("w0", (C, 3, 1, 1), 0.5), _g("b0", (C,), 0.3), relu=True)
    taps, ws = [net.output_nchw(x)], []
    for b in range(nblocks):
        wr, br = _g("wr%d" % b, (mid, C, 1, 1), C ** -0.5), _g("br%d" % b, (mid,), 0.2)
        w3, b3 = _g("w3%d" % b, (mid, mid, 3, 3), (9 * mid) ** -0.5), _g("b3%d" % b, (mid,), 0.2)
        we, be = _g("we%d" % b, (C, mid, 1, 1), mid ** -0.5), _g("be%d" % b, (C,), 0.2)
        r = net.conv(x, wr, br, relu=True)
        t3 = net.conv(r, w3, b3, pad=1, relu=True)
        x = net.conv(t3, we, be, relu=True, residual=x)
        taps.append(net.output_nchw(x))
        ws.append((wr, br, w3, b3, we, be))
    net.finalize()
    return net, taps, ws


def _ref_block(x, w):
    wr, br, w3, b3, we, be = (a.double() for a in w)
    r = F.relu(F.conv2d(x, wr, br)).half().double()                     # the kernel keeps r and t in LDS as fp16
    t = F.relu(F.conv2d(r, w3, b3, padding=1)).half().double()
    return F.relu(F.conv2d(t, we, be) + x)


@pytest.mark.parametrize("C,mid,n,h,w", [(256, 64, 4, 128, 128), (512, 128, 4, 64, 128), (256, 64, 1, 256, 256),
                                          (256, 64, 4, 120, 136), (512, 128, 4, 68, 136), (256, 64, 3, 171, 250)])     # ragged maps: the last patch row / column hangs over the image
def test_fused_bottleneck(cuda_device, C, mid, n, h, w, monkeypatch):
    net, taps, ws = _block_net(cuda_device, C, mid, nblocks=2)
    x = synth.synth_input(5, (n, 3, h, w))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    variants = [v for k, v, ms, fl in net.profile() if k == 1]
    assert variants.count(935000 + C) == 2, variants                  # both blocks ran as one launch each
    xin = outs[taps[0]].double().cpu()
    ref1 = _ref_block(xin, ws[0])
    got1 = outs[taps[1]].double().cpu()
    err1 = float((got1 - ref1).abs().max() / ref1.abs().max())
    ref2 = _ref_block(got1, ws[1])                                      # second block on the first one's actual (fp16) output
    got2 = outs[taps[2]].double().cpu()
    err2 = float((got2 - ref2).abs().max() / ref2.abs().max())
    print("fused bottleneck C %d mid %d: %.2e, %.2e of fp64 (fp16 output rounding 4.9e-4)" % (C, mid, err1, err2))
    assert err1 < 1.5e-3 and err2 < 1.5e-3, (err1, err2)
    # the image border rows / columns, where the r halo is zero padding
    for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), h - 1), (slice(None), slice(None), slice(None), 0),
               (slice(None), slice(None), slice(None), w - 1)):
        assert float((got1[sl] - ref1[sl]).abs().max() / ref1.abs().max()) < 1.5e-3
    assert torch.equal(outs[taps[2]], net.forward(x.to(cuda_device))[taps[2]])       # deterministic
    # against the layer-by-layer kernels of the same build (another summation order and one more rounding point: not bitwise)
    monkeypatch.setenv("GDT_CONV_BNECK", "0")                           # read when a net plans a geometry
    net2, taps2, _ = _block_net(cuda_device, C, mid, nblocks=2)
    net2.set_profiling(True)
    outs2 = net2.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    assert (935000 + C) not in [v for k, v, ms, fl in net2.profile() if k == 1]
    d = float((outs2[taps2[2]].double() - outs[taps[2]].double()).abs().max() / ref2.abs().max())
    assert d < 2e-3, d


@pytest.mark.parametrize("shape", [(4, 128, 128), (4, 120, 136)], ids=["whole-patches", "ragged"])
@pytest.mark.parametrize("projection_first", [True, False])
def test_fused_bottleneck_projection_shortcut(cuda_device, projection_first, shape):
    """layer1's first block: 64 -> (64, 64) -> 256 with a 1x1 projection of the input as the shortcut (torchvision `downsample`); the
    projection is emitted before the reduce conv by engine.py (either order is recognised) -- one launch, against fp64"""
    C, cin, mid = 256, 64, 64
    n, h, w = shape
    net = HipNet(cuda_device, "f16")
    t = net.input(3)
    x = net.conv(t, _g("w0", (cin, 3, 1, 1), 0.5), _g("b0", (cin,), 0.3), relu=True)
    wr, br = _g("wr", (mid, cin, 1, 1), cin ** -0.5), _g("br", (mid,), 0.2)
    wd, bd = _g("wd", (C, cin, 1, 1), cin ** -0.5), _g("bd", (C,), 0.2)
    w3, b3 = _g("w3", (mid, mid, 3, 3), (9 * mid) ** -0.5), _g("b3", (mid,), 0.2)
    we, be = _g("we", (C, mid, 1, 1), mid ** -0.5), _g("be", (C,), 0.2)
    if projection_first:
        sc = net.conv(x, wd, bd)
        r = net.conv(x, wr, br, relu=True)
    else:
        r = net.conv(x, wr, br, relu=True)
        sc = net.conv(x, wd, bd)
    t3 = net.conv(r, w3, b3, pad=1, relu=True)
    y = net.conv(t3, we, be, relu=True, residual=sc)
    taps = [net.output_nchw(x), net.output_nchw(y)]
    net.finalize()
    xi = synth.synth_input(6, (n, 3, h, w))
    net.set_profiling(True)
    outs = net.forward(xi.to(cuda_device))
    torch.cuda.synchronize()
    variants = [v for k, v, ms, fl in net.profile() if k == 1]
    assert 935000 + C + 1 in variants, variants
    xin = outs[taps[0]].double().cpu()
    rr = F.relu(F.conv2d(xin, wr.double(), br.double())).half().double()
    tt = F.relu(F.conv2d(rr, w3.double(), b3.double(), padding=1)).half().double()
    ref = F.relu(F.conv2d(tt, we.double(), be.double()) + F.conv2d(xin, wd.double(), bd.double()))
    got = outs[taps[1]].double().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    print("fused bottleneck with projection shortcut: %.2e of fp64" % err)
    assert err < 1.5e-3, err
    assert torch.equal(outs[taps[1]], net.forward(xi.to(cuda_device))[taps[1]])


@pytest.mark.parametrize("stride,cin,mid,C,n,h,w", [(2, 256, 128, 512, 2, 64, 64), (2, 512, 256, 1024, 1, 64, 64), (1, 512, 128, 512, 1, 40, 52)])
def test_projection_shortcut_folded_into_the_expand_conv(cuda_device, stride, cin, mid, C, n, h, w):
    """first block of a ResNet stage (torchvision `downsample`, stride on the 3x3 conv): where the block does not run as one launch, the
    1x1 projection of the input is accumulated by the expand conv itself -- ONE GEMM over the K-concatenated weights [W_e | W_d], the
    second operand read at the stride-2 pixels of x (conv1x1_rb.hip, CAT form; the projected tensor is never written).  Against fp64 on
    the same fp16 tensors, ragged M (40 x 52 at batch 1), and that this kernel is the one that ran."""
    net = HipNet(cuda_device, "f16")
    t = net.input(3)
    x = net.conv(t, _g("w0", (cin, 3, 1, 1), 0.5), _g("b0", (cin,), 0.3), relu=True)
    wr, br = _g("wr", (mid, cin, 1, 1), cin ** -0.5), _g("br", (mid,), 0.2)
    wd, bd = _g("wd", (C, cin, 1, 1), cin ** -0.5), _g("bd", (C,), 0.2)
    w3, b3 = _g("w3", (mid, mid, 3, 3), (9 * mid) ** -0.5), _g("b3", (mid,), 0.2)
    we, be = _g("we", (C, mid, 1, 1), mid ** -0.5), _g("be", (C,), 0.2)
    sc = net.conv(x, wd, bd, stride=stride)
    r = net.conv(x, wr, br, relu=True)
    t3 = net.conv(r, w3, b3, stride=stride, pad=1, relu=True)
    y = net.conv(t3, we, be, relu=True, residual=sc)
    taps = [net.output_nchw(x), net.output_nchw(t3), net.output_nchw(y)]
    net.finalize()
    xi = synth.synth_input(7, (n, 3, h, w))
    net.set_profiling(True)
    outs = net.forward(xi.to(cuda_device))
    torch.cuda.synchronize()
    variants = [v for k, v, ms, fl in net.profile() if k == 1]
    assert 946128 in variants, variants
    xin, t3v = outs[taps[0]].double().cpu(), outs[taps[1]].double().cpu()
    ref = F.relu(F.conv2d(t3v, we.double(), be.double()) + F.conv2d(xin, wd.double(), bd.double(), stride=stride))
    got = outs[taps[2]].double().cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    print("expand conv + projection shortcut (stride %d): %.2e of fp64 (fp16 output rounding 4.9e-4)" % (stride, err))
    assert err < 1e-3, err
    assert torch.equal(outs[taps[2]], net.forward(xi.to(cuda_device))[taps[2]])


@pytest.mark.parametrize("n,h,w", [(16, 64, 64), (16, 62, 64), (18, 64, 60), (29, 46, 46)], ids=["whole-patches", "ragged-rows", "ragged-columns", "ragged-both-724px-level"])
def test_fused_3x3_expand_layer3(cuda_device, n, h, w, monkeypatch):
    """ResNet-101 layer3 geometry (C 1024, MID 256: the block does not fit conv_bneck's LDS plan): the 3x3 conv and the expand conv + residual run as ONE launch
    (conv3x3_expand_rb.hip, variant 939000 + C / 8; the 256-channel tensor between them stays in LDS), and -- round 5 -- the NEXT block's reduce conv runs as a third
    phase of that launch (variant 938000 + C / 8: `imageretrievalnet.py:189-190`'s Bottleneck chain) -- against fp64 on the same fp16-rounded input with r and t
    rounded where the kernels round them, on the image borders, against the unchained and the layer-by-layer kernels of the same build, deterministic."""
    C, mid = 1024, 256
    net, taps, ws = _block_net(cuda_device, C, mid, nblocks=3)
    x = synth.synth_input(6, (n, 3, h, w))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    variants = [v for k, v, ms, fl in net.profile() if k == 1]
    # three blocks: reduce conv of block 1 on its own, then [3x3 + expand + reduce of block 2], [3x3 + expand + reduce of block 3], [3x3 + expand]
    assert variants.count(938000 + C // 8) == 2 and variants.count(939000 + C // 8) == 1, variants
    assert variants.count(945128) == 1, variants                           # ONE separate 1x1 launch (block 1's reduce) besides the input conv
    prev = outs[taps[0]].double().cpu()
    refs, gots = [], []
    for b in range(3):
        ref = _ref_block(prev, ws[b])                                       # (every block on the previous block's actual fp16 output)
        got = outs[taps[b + 1]].double().cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        print("3x3 + expand (+ next reduce) in one launch, block %d, C %d mid %d, %d x %d x %d: %.2e of fp64 (fp16 output rounding 4.9e-4)" % (b, C, mid, n, h, w, err))
        assert err < 1.5e-3, (b, err)
        for sl in ((slice(None), slice(None), 0), (slice(None), slice(None), h - 1), (slice(None), slice(None), slice(None), 0),
                   (slice(None), slice(None), slice(None), w - 1)):
            assert float((got[sl] - ref[sl]).abs().max() / ref.abs().max()) < 1.5e-3
        refs.append(ref); gots.append(got); prev = got
    assert torch.equal(outs[taps[3]], net.forward(x.to(cuda_device))[taps[3]])       # deterministic
    # the chained reduce against its own launch: the same K order on the same MFMA shape -- bit for bit
    monkeypatch.setenv("GDT_XEXP_CHAIN", "0")                           # read when a net plans a geometry
    net1, taps1, _ = _block_net(cuda_device, C, mid, nblocks=3)
    net1.set_profiling(True)
    outs1 = net1.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    v1 = [v for k, v, ms, fl in net1.profile() if k == 1]
    assert v1.count(939000 + C // 8) == 3 and (938000 + C // 8) not in v1, v1
    assert torch.equal(outs1[taps1[3]], outs[taps[3]])
    monkeypatch.setenv("GDT_CONV_XEXP", "0")
    net2, taps2, _ = _block_net(cuda_device, C, mid, nblocks=3)
    net2.set_profiling(True)
    outs2 = net2.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    v2 = [v for k, v, ms, fl in net2.profile() if k == 1]
    assert (939000 + C // 8) not in v2 and (938000 + C // 8) not in v2
    d = float((outs2[taps2[3]].double() - outs[taps[3]].double()).abs().max() / refs[2].abs().max())
    assert d < 3e-3, d
