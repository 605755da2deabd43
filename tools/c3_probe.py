"""dev: where BASELINE config 4's per-rank workload (8 x 3 x 1024^2, hub-default pyramid, GeM-ResNet-101 + whitening) spends its time: each pyramid level alone
(device time of its forward, and its wall time per call = host launch time when that is larger), the three levels one after the other / concurrent, the per-op
table of each level, and the same with the levels' forwards replayed from hipGraphs"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import _c3_network
from gandtr_amd import engine
from gandtr_amd.tools import synth
import bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def wall(fn, k=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def host_only(fn, k=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t = (time.perf_counter() - t0) / k * 1e3
    torch.cuda.synchronize()
    return t


with torch.no_grad():
    sd = synth.resnet101_state(0)
    net = engine.build_embedder(sd, dev)
    net.use_graphs = False
    x = synth.synth_input(5, (n, 3, 1024, 1024)).to(dev)
    scales = [1.0, 2 ** -0.5, 0.5]
    tot = 0.0
    for s in scales:
        t = wall(lambda: net.forward(x, scale=s))
        h = host_only(lambda: net.forward(x, scale=s))
        net.set_profiling(True); net.forward(x, scale=s); torch.cuda.synchronize()
        prof = net.profile(); net.set_profiling(False)
        dev_ms = sum(ms for k, v, ms, fl in prof)
        per = {}
        for k, v, ms, fl in prof:
            if k == 1 and ms > 0:
                e = per.setdefault(bench.kernel_name(v), [0.0, 0]); e[0] += ms; e[1] += 1
        tot += t
        print("scale %.3f: wall %.2f ms, host issue %.2f ms, sum of op times %.2f ms" % (s, t, h, dev_ms))
        print("   " + ", ".join("%s %.2f/%d" % (k[:34], e[0], e[1]) for k, e in sorted(per.items(), key=lambda kv: -kv[1][0])[:8]))
    print("levels one by one: %.2f ms = %.0f desc/s" % (tot, n / tot * 1e3))
    t = wall(lambda: net.forward_many([(x, s) for s in scales]))
    print("forward_many (side streams): %.2f ms = %.0f desc/s" % (t, n / t * 1e3))
    net.use_graphs = True
    for s in scales:
        for _ in range(3): net.forward(x, scale=s)
    t = wall(lambda: [net.forward(x, scale=s) for s in scales])
    print("levels one by one, hipGraph replay: %.2f ms = %.0f desc/s" % (t, n / t * 1e3))
with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
    hub = _c3_network(dev, True, tmp)
    t = wall(lambda: hub(x))
    print("hub network (wrappers, forward_many, aggregate, whiten): %.2f ms = %.0f desc/s" % (t, n / t * 1e3))
