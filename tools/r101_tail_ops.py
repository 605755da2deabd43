import os, sys
sys.path.insert(0, "/root/repo")
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
x = synth.synth_input(1, (32, 3, 1024, 1024)).to(dev)
for _ in range(3): net.forward(x)
net.set_profiling(True)
net.forward(x); torch.cuda.synchronize()
rows = list(net.profile()); by = net.profile_bytes()
for i, ((kind, tile, ms, fl), b) in enumerate(zip(rows, by)):
    if ms > 0.01 and i >= 95: print("%3d kind %d variant %7d  %7.3f ms  %7.1f TF  %7.1f GB/s" % (i, kind, tile, ms, fl / ms / 1e9 if ms > 0 else 0, b / ms / 1e6))
