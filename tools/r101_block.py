"""Developer tool: one ResNet-101 layer3 Bottleneck (1024 -> 256 -> 256 -> 1024 + residual) at 32 x 64 x 64, per-op timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = HipNet(dev)
t = net.input(3)
x = net.conv(t, synth._normal(0, "w0", (1024, 3, 1, 1), 0.5), relu=True)
for b in range(3):
    o = net.conv(x, synth._normal(0, "a%d" % b, (256, 1024, 1, 1), 0.03), relu=True)
    o = net.conv(o, synth._normal(0, "b%d" % b, (256, 256, 3, 3), 0.02), pad=1, relu=True)
    x = net.conv(o, synth._normal(0, "c%d" % b, (1024, 256, 1, 1), 0.03), relu=True, residual=x)
net.out = net.gem_l2n(x, 3.0)
net.finalize()
xin = synth.synth_input(1, (32, 3, 64, 64)).to(dev)
for _ in range(3): net.forward(xin)
net.set_profiling(True)
acc = None
for _ in range(5):
    net.forward(xin); torch.cuda.synchronize()
    p = net.profile()
    acc = p if acc is None else [(a[0], a[1], a[2] + b[2], a[3]) for a, b in zip(acc, p)]
for i, (k, v, ms, fl) in enumerate(acc):
    if k == 1: print("%2d var %6d %.3f ms %7.1f TF" % (i, v, ms / 5, fl / (ms / 5) / 1e9))
