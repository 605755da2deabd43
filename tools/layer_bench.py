"""dev tool: sustained timing of ONE conv layer through the C ABI (power-throttled steady state, unlike the per-op event
profile of tools/dump_ops.py).  usage: tools/layer_bench.py cin cout k stride H W N [res] [relu] [iters]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gandtr_amd.engine import HipNet
from gandtr_amd.tools import synth

cin, cout, k, stride, H, W, N = (int(v) for v in sys.argv[1:8])
res = len(sys.argv) > 8 and sys.argv[8] == "1"
relu = len(sys.argv) > 9 and sys.argv[9] == "1"
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 100
dev = torch.device("cuda:0")
net = HipNet(dev)
t = net.input(3)
a = net.conv(t, synth._normal(0, "w0", (cin, 3, 1, 1), 0.5))
r = net.conv(t, synth._normal(0, "w1", (cout, 3, 1, 1), 0.5), stride=stride) if res else -1
# a chain of `depth` identical layers would change shapes; instead the layer under test is repeated on the same input
outs = []
for i in range(8):
    o = net.conv(a, synth._normal(0, "w", (cout, cin, k, k), 0.02), pad=k // 2, stride=stride, relu=relu, residual=r)
    outs.append(o)
g = net.gem_l2n(outs[-1], 3.0)
net.finalize()
x = synth.synth_input(1, (N, 3, H, W)).to(dev)
for _ in range(3): net.forward(x)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters): net.forward(x)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
net.set_profiling(True); net.forward(x); p = net.profile()
other = sum(ms for kind, var, ms, fl in p if not (kind == 1 and fl > 0 and var != p[1][1])) if False else 0
convs = [(var, ms, fl) for kind, var, ms, fl in p if kind == 1][-8:]
fl = convs[-1][2]
print("layer %dx%d %d->%d s%d @%dx%d N=%d res=%d: sustained %.3f ms per forward (8 layers + lift + gem), variant %d, profiled %.3f ms/layer, %.1f GFLOP/layer"
      % (k, k, cin, cout, stride, H, W, N, res, dt * 1e3, convs[-1][0], sum(c[1] for c in convs) / 8, fl / 1e9))
