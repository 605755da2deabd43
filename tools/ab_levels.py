"""dev: pyramid levels as joined launches (gdt_net_forward_levels) vs one side stream per level, alternating inside one process; engine level and hub level"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import _c3_network
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
preset = sys.argv[2] if len(sys.argv) > 2 else "hub"
scales = [1.0, 2 ** -0.5, 0.5] if preset == "hub" else [1.0, 2 ** -0.5, 2 ** 0.5]


def wall(fn, k=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


with torch.no_grad(), tempfile.TemporaryDirectory() as tmp:
    net = engine.build_embedder(synth.resnet101_state(0), dev)
    x = synth.synth_input(5, (n, 3, 1024, 1024)).to(dev)
    hub = _c3_network(dev, True if preset == "hub" else "sms", tmp)
    for rep in range(3):
        for mode in ("1", "0"):
            os.environ["GANDTR_HIP_JOINT_LEVELS"] = mode
            t = wall(lambda: net.forward_many([(x, s) for s in scales]))
            th = wall(lambda: hub(x))
            print("joint %s: engine %.2f ms = %.0f desc/s; hub %.2f ms = %.0f desc/s" % (mode, t, n / t * 1e3, th, n / th * 1e3))
