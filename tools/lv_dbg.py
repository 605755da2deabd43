import sys, torch
sys.path.insert(0, '/root/repo')
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
size = int(sys.argv[1])
x = synth.synth_input(91, (8, 3, size, size)).to(dev)
net.forward_many([(x, 1.0), (x, 2 ** -0.5), (x, 0.5)])
torch.cuda.synchronize()
print("joined", net.levels_joined())
