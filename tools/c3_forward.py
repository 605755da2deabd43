"""dev / profiling: N calls of BASELINE config 4's per-rank workload (8 x 3 x 1024 x 1024 through the hub-style multi-scale + whitening GeM-ResNet-101), nothing else --
the command behind profiles/r05_c3_kernel_stats.csv.  usage: python tools/c3_forward.py [calls] [hub|sms]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import _c3_network
from gandtr_amd.tools import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
preset = sys.argv[2] if len(sys.argv) > 2 else "hub"
dev = torch.device("cuda:0")
with torch.no_grad(), tempfile.TemporaryDirectory() as tmp:
    net = _c3_network(dev, True if preset == "hub" else "sms", tmp)
    x = synth.synth_input(4, (8, 3, 1024, 1024)).to(dev)
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
print("done", n)
