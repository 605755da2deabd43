"""dev: generator at batch 1 / 2 / 4, default f16c mode: eager launches vs hipGraph replay (HipNet.use_graphs), back to back and per synchronised call; per-op table at batch 1"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")


def thr(fn, n=40, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def lat(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3


sd = synth.generator_state(0, "instance")
net = engine.build_generator(sd, dev)
for shape in ((1, 3, 256, 256), (2, 3, 256, 256), (4, 3, 256, 256)):
    x = synth.synth_input(1, shape, 1.0).to(dev)
    net.use_graphs = False
    e = (thr(lambda: net.forward(x)), lat(lambda: net.forward(x)))
    net.use_graphs = True
    for _ in range(3): net.forward(x)
    g = (thr(lambda: net.forward(x)), lat(lambda: net.forward(x)))
    print("%s eager %.3f ms back to back / %.3f ms per synchronised call; graph replay %.3f / %.3f" % (shape, e[0], e[1], g[0], g[1]))
net.use_graphs = False
x = synth.synth_input(1, (1, 3, 256, 256), 1.0).to(dev)
net.set_profiling(True); net.forward(x); torch.cuda.synchronize()
per = {}
tot = 0.0
for k, v, ms, fl in net.profile():
    tot += ms
    if ms > 0:
        name = bench.kernel_name(v) if k == 1 else {0: "input", 2: "inorm", 5: "tap"}.get(k, str(k))
        e = per.setdefault(name, [0.0, 0]); e[0] += ms; e[1] += 1
print("batch 1 per-op (events around every op): total %.3f ms" % tot)
for name, e in sorted(per.items(), key=lambda kv: -kv[1][0]): print("   %-46s %.3f ms / %d" % (name, e[0], e[1]))
