import os, sys, time, collections
sys.path.insert(0, os.getcwd())
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.vgg16_state(0), dev)
for shape in [(32, 3, 1024, 683), (32, 3, 1024, 768)]:
    x = synth.synth_input(1, shape).to(dev)
    for _ in range(3): net.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): net.forward(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(shape, "%.3f ms  %.0f desc/s  %.2f ns/pixel" % (dt * 1e3, shape[0] / dt, dt * 1e9 / (shape[0] * shape[2] * shape[3])))
    net.set_profiling(True); net.forward(x); torch.cuda.synchronize()
    for i, (kind, tile, ms, fl) in enumerate(net.profile()):
        if ms > 0.2: print("   %2d kind %d variant %7d %7.3f ms %7.1f TF" % (i, kind, tile, ms, fl / ms / 1e9 if ms else 0))
    net.set_profiling(False)
