"""dev: per-op time of one f16c BatchNorm (hedngan) generator forward (64 x 256^2) from the executor's event profile"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_generator(synth.generator_state(0, "batch"), dev, precision="f16c")
x = synth.synth_input(1, (64, 3, 256, 256)).to(dev)
for _ in range(3): net.forward(x)
net.set_profiling(True)
net.forward(x); torch.cuda.synchronize()
tot = 0.0
for i, (kind, tile, ms, fl) in enumerate(net.profile()):
    tot += ms
    print("%3d kind %d variant %7d  %7.3f ms  %7.1f TF" % (i, kind, tile, ms, fl / ms / 1e9 if ms > 0 else 0))
print("total %.3f ms" % tot)
