"""dev tool: throughput of the ingest launches (gandtr_amd/csrc/ingest.hip) against the HBM roofline, Pillow + numpy timed beside.
usage: tools/ingest_bench.py [H W imsize] [iters] [--clahe]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from gandtr_amd import ingest

args = [a for a in sys.argv[1:] if not a.startswith("--")]
h, w, s = (int(v) for v in args[:3]) if len(args) >= 3 else (1200, 1600, 1024)
iters = int(args[3]) if len(args) > 3 else 200
clip = 1.0 if "--clahe" in sys.argv else None
dev = torch.device("cuda:0")
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
a = np.stack([127 + 100 * np.sin(xx / 90) * np.cos(yy / 70), 127 + 90 * np.cos(xx / 50 + yy / 110), 60 + 0.1 * xx], -1)
a = np.clip(a + np.random.default_rng(1).normal(0, 6, a.shape), 0, 255).astype(np.uint8)
x = torch.from_numpy(a).to(dev)
mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
for _ in range(5):
    y = ingest.ingest(x, s, mean, std, clahe_clip=clip)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = ingest.ingest(x, s, mean, std, clahe_clip=clip)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
size = ingest.thumbnail_size(w, h, s) or (w, h)
fx, fy, _ = ingest.reduce_plan(w, h, *size)
rh, rw = -(-h // fy), -(-w // fx)
# algorithmic bytes: source once, reduced image written + read (if any), 8-bit intermediate written + read, fp32 CHW written
alg = h * w * 3 + (2 * rh * rw * 3 if (fx > 1 or fy > 1) else 0) + 2 * rh * size[0] * 3 + size[0] * size[1] * 12
if clip is not None:
    alg += size[0] * size[1] * (12 + 38)                # [0,1] planes read back + the CLAHE launches (clahe.hip: 38 B / pixel)
out = {"workload": "ingest %dx%dx3 u8 -> thumbnail %d%s -> fp32 CHW %dx%d" % (h, w, s, " + clahe" if clip else "", size[1], size[0]),
       "ms": round(ms, 4), "images_per_s": round(1e3 / ms, 1), "reduce": [fx, fy],
       "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / ms / 1e6 / 8000.0, 4),
                    "algorithmic_bytes": alg}}
try:
    from PIL import Image
    t0 = time.time(); n = 0
    while time.time() - t0 < 3.0:
        p = Image.fromarray(a); p.thumbnail((s, s), Image.LANCZOS)
        t = (np.asarray(p).astype(np.float32) / 255.0).transpose(2, 0, 1)
        t = (t - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
        n += 1
    out["cpu_baseline"] = {"value": round(n / (time.time() - t0), 2), "unit": "images/s", "cores": 1, "kind": "reference",
                           "sample": "Pillow thumbnail + numpy totensor/normalize (no CLAHE), %d images" % n}
except ImportError:
    pass
# a mixed-size set on concurrent streams (what a dataset pass looks like)
rng = np.random.default_rng(3)
sizes = [(1200, 1600), (1600, 1200), (1024, 768), (768, 1024), (1333, 2000), (2000, 1333), (960, 1280), (1500, 1500)] * 8
photos = [torch.from_numpy(rng.integers(0, 256, (hh, ww, 3)).astype(np.uint8)).to(dev) for hh, ww in sizes]
for nstreams in (1, 4, 8):
    for _ in range(2):
        ingest.ingest_many(photos, s, mean, std, clahe_clip=clip, streams=nstreams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ingest.ingest_many(photos, s, mean, std, clahe_clip=clip, streams=nstreams)
    torch.cuda.synchronize()
    out["mixed_64_images_per_s_streams_%d" % nstreams] = round(5 * len(photos) / (time.perf_counter() - t0), 1)
print(json.dumps(out))
