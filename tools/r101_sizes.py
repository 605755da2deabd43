"""dev: GeM-ResNet-101 forward time at the sizes of the multi-scale pyramid (batch 32), with / without the fused layer3 launch (run twice: GDT_CONV_XEXP=0 / 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
for shape in [(32, 3, 1024, 1024), (32, 3, 724, 724), (32, 3, 512, 512), (32, 3, 1024, 768), (32, 3, 1024, 683), (16, 3, 1024, 1024), (24, 3, 1024, 1024)]:
    x = synth.synth_input(1, shape).to(dev)
    for _ in range(3): net.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net.forward(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(shape, "%.3f ms  %.0f desc/s" % (ms, shape[0] / ms * 1e3))
