"""dev tool: throughput of the CLAHE post-processing launches (gandtr_amd/csrc/clahe.hip) against the HBM roofline.
usage: tools/clahe_bench.py [N H W] [iters]   -- prints one JSON line; algorithmic bytes = 38 B / pixel (clahe.hip header)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from gandtr_amd import clahe

n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 256, 256)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
low = torch.nn.functional.interpolate(torch.randn(n, 3, 9, 9, generator=g), size=(h, w), mode="bicubic", align_corners=False)
x = torch.tanh(0.9 * low + 0.15 * torch.randn(n, 3, h, w, generator=g)).to(dev)
pair = ([0.5] * 3, [0.5] * 3)
for _ in range(5):
    y = clahe.clahe_lab(x, 1.0, 8, pair, pair)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = clahe.clahe_lab(x, 1.0, 8, pair, pair)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
px = n * h * w
out = {"workload": "clahe_post %dx3x%dx%d clip 1.0 grid 8" % (n, h, w), "ms": round(ms, 4), "images_per_s": round(n / ms * 1e3, 1),
       "roofline": {"bound": "hbm", "achieved": round(38 * px / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(38 * px / ms / 1e6 / 8000.0, 4)}}
if "--cpu" in sys.argv:
    from oracle import clahe_oracle as C
    k = min(n, 8)
    xs = x[:k].cpu().numpy()
    t0 = time.time(); C.clahe_post(xs, *pair, 1.0); dt = time.time() - t0
    out["cpu_baseline"] = {"value": round(k / dt, 2), "unit": "images/s", "cores": 1, "kind": "port", "sample": "%d images" % k}
print(json.dumps(out))
