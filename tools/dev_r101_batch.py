"""dev: GeM-ResNet-101 @1024 descriptors/s as a function of the batch size (is the batch-32 forward paying for a working set that
falls out of the 256 MB Infinity Cache?)"""
import sys, time, torch
sys.path.insert(0, ".")
from gandtr_amd import engine
from gandtr_amd.tools import synth

dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
for n in (32, 16, 8, 4, 2, 32):
    x = synth.synth_input(1, (n, 3, 1024, 1024)).to(dev)
    for _ in range(2):
        net.forward(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = max(3, 96 // n)
    for _ in range(reps):
        net.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("batch %2d: %.2f ms  %.0f desc/s" % (n, dt * 1e3, n / dt), flush=True)
