"""Developer tool: per-op timing table (HIP events inside the library) for the generator / embedders on one GPU."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth

which = sys.argv[1] if len(sys.argv) > 1 else "gen"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
dev = torch.device("cuda:0")
if which == "gen":
    net = engine.build_generator(synth.generator_state(0, "instance"), dev, precision=prec); x = synth.synth_input(1, (64, 3, 256, 256), 1.0).to(dev)
elif which == "genbn":
    net = engine.build_generator(synth.generator_state(0, "batch"), dev, precision=prec); x = synth.synth_input(1, (64, 3, 256, 256), 1.0).to(dev)
elif which == "hed":
    net = engine.build_hed(synth.hed_state(0), dev, perm=[2, 1, 0], in_affine=([0.5] * 3, [0.09212946, 0.04247542, 0.01890622])); x = synth.synth_input(1, (64, 3, 256, 256), 1.0).to(dev)
elif which == "r101":
    net = engine.build_embedder(synth.resnet101_state(0), dev, precision=prec); x = synth.synth_input(1, (32, 3, 1024, 1024)).to(dev)
else:
    net = engine.build_embedder(synth.vgg16_state(0), dev, precision=prec); x = synth.synth_input(1, (8, 3, 1024, 1024)).to(dev)
for _ in range(3): net.forward(x)
net.set_profiling(True)
acc = None
for _ in range(5):
    net.forward(x); torch.cuda.synchronize()
    p = net.profile()
    acc = p if acc is None else [(a[0], a[1], a[2] + b[2], a[3]) for a, b in zip(acc, p)]
tot = sum(a[2] for a in acc) / 5
names = {0: "input", 1: "conv", 2: "inorm", 3: "maxpool", 4: "gem", 5: "tap", 6: "hed"}
for i, (k, t, ms, fl) in enumerate(acc):
    ms /= 5
    print("%3d %-8s var %6d  %8.3f ms  %6.1f%%  %8.2f GFLOP  %7.1f TFLOP/s" % (i, names[k], t, ms, 100 * ms / tot, fl / 1e9, fl / ms / 1e9 if ms > 0 else 0))
print("total %.3f ms" % tot)
