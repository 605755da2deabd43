"""dev: cost of the residual + write-back staging modes of conv3x3_halo_c (MODE 1 / 3 / 7) on the resblock shape"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth

dev = torch.device("cuda:0")
g = lambda name, shape, std: synth._normal(0, name, shape, std)
for mode in ("plain IN+ReLU (1)", "IN + residual (3)", "IN + residual + write-back (7)"):
    net = HipNet(dev, "f16c")
    t = net.input(3)
    t0 = net.conv(t, g("w0", (256, 3, 1, 1), 0.7))
    r = net.conv(t, g("w1", (256, 3, 1, 1), 0.7))
    outs = []
    for k in range(6):
        if mode.startswith("plain"):
            tn = net.instance_norm(t0, relu=True)
        else:
            tn = net.instance_norm(t0, relu=False, residual=r)
        o = net.conv(tn, g("w", (256, 256, 3, 3), 0.02), g("b", (256,), 0.1), pad=1, reflect=True)
        outs.append(o)
        if mode.endswith("(7)"):
            outs.append(net.conv(tn, g("w2", (256, 256, 3, 3), 0.02), g("b", (256,), 0.1), pad=1, reflect=True))   # second consumer: materialise
    net.gem_l2n(outs[-1], 3.0)
    net.finalize()
    x = synth.synth_input(1, (64, 3, 64, 64)).to(dev)
    for _ in range(3):
        net.forward(x)
    net.set_profiling(True)
    acc = {}
    for _ in range(5):
        net.forward(x); torch.cuda.synchronize()
        for i, (k, v, ms, fl) in enumerate(net.profile()):
            acc[i] = (k, v, acc.get(i, (0, 0, 0.0))[2] + ms / 5)
    print(mode, " ".join("%d:%.3f" % (v, ms) for k, v, ms in acc.values() if k == 1 and v >= 970000))
