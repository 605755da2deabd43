"""dev: the exact-mode (f16x3) generator forward in a loop with the engine clock and socket power sampled beside it (bench.py's ClockSampler)
usage: python tools/x3_clock.py [seconds] [f16x3|f16c|f16|r101|vgg16]   (GANDTR_HIP_LIB selects a variant library)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import ClockSampler
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
if prec in ("r101", "vgg16"):            # the embedders (fp16 mode), 32 x 1024^2
    net = engine.build_embedder(synth.resnet101_state(0) if prec == "r101" else synth.vgg16_state(0), dev)
    x = synth.synth_input(3, (32, 3, 1024, 1024)).to(dev)
else:
    net = engine.build_generator(synth.generator_state(0, "instance", gain=0.02), dev, precision=prec)
    x = synth.synth_input(3, (64, 3, 256, 256), 1.0).to(dev)
with torch.no_grad():
    for _ in range(5): net.forward(x)
    torch.cuda.synchronize()
    n = 0
    with ClockSampler(dev) as cs:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for _ in range(5): net.forward(x)
            torch.cuda.synchronize(); n += 5
        dt = time.perf_counter() - t0
    print("%s: %.3f ms per %d images = %.0f images/s; clocks %s" % (prec, dt / n * 1e3, x.shape[0], x.shape[0] * n / dt, cs.summary()))
