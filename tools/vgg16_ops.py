"""dev: per-op time of one GeM-VGG16 forward (32 x 1024^2) from the executor's own event profile: op index, kind, kernel variant, ms, TFLOP/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.vgg16_state(0), dev)
x = synth.synth_input(1, (32, 3, 1024, 1024)).to(dev)
for _ in range(3): net.forward(x)
net.set_profiling(True)
net.forward(x); torch.cuda.synchronize()
tot = 0.0
for i, (kind, tile, ms, fl) in enumerate(net.profile()):
    tot += ms
    if ms > 0.05: print("%3d kind %d variant %7d  %7.3f ms  %7.1f TF" % (i, kind, tile, ms, fl / ms / 1e9 if ms > 0 else 0))
print("total %.3f ms" % tot)
