"""dev: per-kernel-variant time of GeM-ResNet-101 forwards at the small geometries of configs 4 / 5 (event profile)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
for shape in [(1, 3, 1024, 1024), (8, 3, 512, 512), (8, 3, 724, 724), (128, 3, 256, 256), (32, 3, 1024, 683)]:
    x = synth.synth_input(1, shape).to(dev)
    for _ in range(3): net.forward(x)
    net.set_profiling(True)
    net.forward(x); torch.cuda.synchronize()
    acc = collections.OrderedDict(); tot = 0.0
    for kind, tile, ms, fl in net.profile():
        key = (kind, tile); a = acc.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl; tot += ms
    print(shape, "total %.3f ms" % tot)
    for (kind, tile), (n, ms, fl) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   kind %d variant %7d  x%3d  %6.3f ms  %6.1f TF" % (kind, tile, n, ms, fl / ms / 1e9 if ms > 0 else 0))
    net.set_profiling(False)
