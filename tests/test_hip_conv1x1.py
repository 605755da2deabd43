"""GPU parity of the streaming 1x1 conv kernel (conv1x1_rb.hip: the ResNet-101 Bottleneck reduce / expand convs) against an fp64
evaluation of the same layer on the same fp16-rounded input: single, even and deep K-loops, the residual + ReLU epilogue, ragged M
(rows not a multiple of the 128-row tile), and that the kernel is the one that ran."""
import pytest
import torch
import torch.nn.functional as F

from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth

pytestmark = pytest.mark.gpu


def _g(name, shape, std):
    return synth._normal(0, name, shape, std)


@pytest.mark.parametrize("cin,cout,hw,n,res,relu", [
    (64, 256, 96, 4, True, True),          # one K-step per tile (layer1 expand)
    (128, 512, 96, 2, True, True),         # two K-steps
    (256, 1024, 64, 4, True, True),        # layer3 expand + residual + ReLU
    (1024, 256, 64, 8, False, True),       # layer3 reduce: 16 K-steps
    (512, 128, 93, 8, False, False),       # ragged M: 8 * 93 * 93 rows
    (256, 512, 93, 2, True, False),        # ragged M with the residual path
])
def test_conv1x1_streaming(cuda_device, cin, cout, hw, n, res, relu):
    net = HipNet(cuda_device, "f16")
    t = net.input(3)
    a = net.conv(t, _g("w0", (cin, 3, 1, 1), 0.5), relu=True)
    r = net.conv(t, _g("w1", (cout, 3, 1, 1), 0.5)) if res else -1
    w, bias = _g("w", (cout, cin, 1, 1), cin ** -0.5), _g("b", (cout,), 0.2)
    o = net.conv(a, w, bias, relu=relu, residual=r)
    taps = [net.output_nchw(o), net.output_nchw(a)] + ([net.output_nchw(r)] if res else [])
    net.finalize()
    x = synth.synth_input(3, (n, 3, hw, hw))
    net.set_profiling(True)
    outs = net.forward(x.to(cuda_device))
    torch.cuda.synchronize()
    variants = [v for k, v, ms, fl in net.profile() if k == 1]
    assert variants[-1] == 945128, variants                       # the streaming kernel ran the layer under test
    ain = outs[taps[1]].double().cpu()                            # its actual (fp16-rounded) input and residual
    ref = F.conv2d(ain, w.double(), bias.double())
    if res:
        ref = ref + outs[taps[2]].double().cpu()
    if relu:
        ref = F.relu(ref)
    got = outs[taps[0]].double().cpu()
    assert got.shape == ref.shape
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1.5e-3, err                                      # one fp16 rounding of the output (+ of the pre-residual value)
    # run-to-run determinism
    assert torch.equal(outs[taps[0]], net.forward(x.to(cuda_device))[taps[0]])
