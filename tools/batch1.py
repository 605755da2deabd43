"""Developer tool: latency at the reference's own operating point (batch size 1, hub API, per-call host overhead included)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import hubconf
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
def lat(fn, n=50, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    g = hubconf.cyclegan(pretrained=False, device=dev)
    for shape in ((1, 3, 256, 256), (1, 3, 362, 362), (1, 3, 1024, 1024)):
        x = synth.synth_input(1, shape, 1.0).to(dev)
        print("cyclegan", shape, "%.3f ms" % lat(lambda: g(x)))
    for name in ("gem_vgg16_cyclegan", "gem_resnet101_hedngan"):
        e = getattr(hubconf, name)(pretrained=False, device=dev)
        for shape in ((1, 3, 362, 362), (1, 3, 1024, 1024), (1, 3, 1024, 768)):
            x = synth.synth_input(2, shape).to(dev)
            print(name, shape, "%.3f ms" % lat(lambda: e(x), n=20, w=5))
