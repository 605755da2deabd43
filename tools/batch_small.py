"""dev: generator latency at small batches in the default f16c mode vs the opt-in f16 / f16x3 modes (engine API)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
def lat(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
sd = synth.generator_state(0, "instance")
for prec in ("f16c", "f16", "f16x3"):
    net = engine.build_generator(sd, dev, precision=prec)
    for shape in ((1, 3, 256, 256), (2, 3, 256, 256), (4, 3, 256, 256), (8, 3, 256, 256), (16, 3, 256, 256), (1, 3, 1024, 1024)):
        x = synth.synth_input(1, shape, 1.0).to(dev)
        print("%-6s %s %.3f ms  (%.0f images/s)" % (prec, shape, lat(lambda: net.forward(x)), shape[0] / lat(lambda: net.forward(x)) * 1e3), flush=True)
