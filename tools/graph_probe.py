"""dev: the whole multi-scale call (three levels on side streams + aggregation + whitening) captured into ONE hipGraph: per-call latency and back-to-back rate vs eager"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import _c3_network
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def lat(fn, k=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def thr(fn, k=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


with torch.no_grad(), tempfile.TemporaryDirectory() as tmp:
    hub = _c3_network(dev, True, tmp)
    x = synth.synth_input(5, (n, 3, 1024, 1024)).to(dev)
    ref = hub(x).clone()
    print("eager: latency %.2f ms, back to back %.2f ms" % (lat(lambda: hub(x)), thr(lambda: hub(x))))
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(2): hub(x)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = hub(x)
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", bool(torch.equal(out, ref)))
    print("graph: latency %.2f ms, back to back %.2f ms" % (lat(lambda: g.replay()), thr(lambda: g.replay())))
