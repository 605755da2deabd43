"""dev: GeM-ResNet-101 forward time at small geometries (few patch tiles per layer)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
for shape in [(128, 3, 256, 256), (8, 3, 512, 512), (8, 3, 1024, 1024), (1, 3, 1024, 1024), (4, 3, 724, 724)]:
    x = synth.synth_input(1, shape).to(dev)
    for _ in range(3): net.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net.forward(x)
    torch.cuda.synchronize()
    print(shape, "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
