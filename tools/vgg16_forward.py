"""dev / profiling: N forwards of GeM-VGG16 on 32 x 3 x 1024 x 1024 (the secondary headline geometry), nothing else -- the command the
FETCH_SIZE / WRITE_SIZE passes of profiles/rNN_pmc_traffic_r101.json run.  usage: python tools/r101_forward.py [forwards]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gandtr_amd import engine
from gandtr_amd.tools import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.vgg16_state(0), dev)
x = synth.synth_input(1, (32, 3, 1024, 1024)).to(dev)
for _ in range(n):
    net.forward(x)
torch.cuda.synchronize()
print("done", n)
