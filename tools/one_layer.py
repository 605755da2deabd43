import sys, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/gandtr_amd') else os.getcwd())
import torch
from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = HipNet(dev)
t = net.input(3)
t = net.conv(t, synth._normal(0, "w0", (256, 3, 1, 1), 0.5))
for i in range(4):
    t = net.conv(t, synth._normal(0, "w%d" % i, (256, 256, 3, 3), 0.02), pad=1, reflect=True, relu=True)
o = net.gem_l2n(t, 3.0)
net.finalize()
x = synth.synth_input(1, (64, 3, 64, 64)).to(dev)
for _ in range(5): net.forward(x)
torch.cuda.synchronize()
net.set_profiling(True)
acc = None
for _ in range(10):
    net.forward(x)
    p = net.profile()
    ms = [r[2] for r in p]
    acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
print("per-op ms:", " ".join("%s:%.3f" % (p[i][0], acc[i] / 10) for i in range(len(p))))
