"""dev: GeM-ResNet-101 32 x 1024^2 as ONE forward vs TWO half-batch forwards on two streams (HipNet.forward_many), the second delayed by a spin of D microseconds --
do the MFMA-bound kernels of one half overlap the HBM-bound kernels of the other?   usage: python tools/r101_split.py [delays_us ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
net = engine.build_embedder(synth.resnet101_state(0), dev)
x = synth.synth_input(1, (32, 3, 1024, 1024)).to(dev)
parts = int(os.environ.get("PARTS", "2"))
chunks = list(x.chunk(parts))

def timed(fn, steps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3

ms = timed(lambda: net.forward(x))
print("one forward of 32: %.3f ms = %.0f desc/s" % (ms, 32e3 / ms))
ref = net.forward(x)[net.out_slot].clone()
ms = timed(lambda: net.forward_many([(c, None) for c in chunks]))
print("%d forwards of %d on %d streams, no delay: %.3f ms = %.0f desc/s" % (parts, 32 // parts, parts, ms, 32e3 / ms))
got = torch.cat([o[net.out_slot] for o in net.forward_many([(c, None) for c in chunks])])
print("max |d| vs one forward: %.2e" % float((got - ref).abs().max()))
pools = net.__dict__["_side"]
cur = torch.cuda.current_stream(dev)
for delay in [float(a) for a in sys.argv[1:]]:
    def run():
        outs = []
        for k, c in enumerate(chunks):
            st = pools["streams"][k]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                if k: torch.cuda._sleep(int(delay * k * 1800))
                n, _, h, w = c.shape
                o = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in net.output_shapes(n, h, w)]
                net._launch(c, n, h, w, h, w, 1.0, pools["ws"][k], o)
                outs.append(o)
        for k in range(len(chunks)): cur.wait_stream(pools["streams"][k])
        return outs
    ms = timed(run)
    print("delay %5.0f us: %.3f ms = %.0f desc/s" % (delay, ms, 32e3 / ms))
