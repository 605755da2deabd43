"""dev tool: engine clock / power while the MFMA-only micro-benchmark runs (is its 1.7 PFLOP/s a clock limit or an issue limit?)"""
import ctypes, os, sys, json
sys.path.insert(0, os.getcwd())
import torch
import bench
from gandtr_amd import _hip
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = _hip.load()
v = ctypes.c_double(0.0)
_hip.check(lib.gdt_mfma_only_tflops(50, ctypes.byref(v), None))
for ms in (300, 1000):
    s = bench.ClockSampler(dev)
    with s:
        _hip.check(lib.gdt_mfma_only_tflops(ms, ctypes.byref(v), None))
    c = s.summary()
    print(json.dumps({"millis": ms, "tflops": round(v.value, 1), "clocks": c,
                      "implied_util_at_clock": round(v.value / (2500.0 * c["sclk_mhz_mean"] / 2400.0), 3) if c else None}))
