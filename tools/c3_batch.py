"""dev: multi-scale + whitening GeM-ResNet-101 (BASELINE config 4's network) at batch 8 / 16 / 32, pyramid levels concurrent vs level by level"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import _c3_network
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
    net = _c3_network(dev, True, tmp)
    for n in (8, 16, 32):
        x = synth.synth_input(5, (n, 3, 1024, 1024)).to(dev)
        for mode in ("1", "0"):
            os.environ["GANDTR_HIP_CONCURRENT_LEVELS"] = mode
            for _ in range(2): net(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): net(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
            print("batch %2d levels %s: %.2f ms = %.0f desc/s" % (n, "concurrent" if mode == "1" else "one by one", dt * 1e3, n / dt))
