"""dev / measurement: device JPEG decoding against the reference's loader (Pillow on the host), photo-like synthetic files.
usage: python tools/jpeg_bench.py [n_files] [width] [height]   -> one JSON line"""
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from gandtr_amd import ingest, jpeg


def picture(w, h, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    planes = []
    for c in range(3):
        base = 128 + 70 * np.sin(xx / (37.0 + 9 * c) + seed) * np.cos(yy / (51.0 + 5 * c)) + 30 * np.sin((xx + yy) / 123.0)
        texture = rng.normal(0, 10, (h, w)) + 40.0 * (((xx // 61 + yy // 47) % 2) > 0)
        planes.append(np.clip(base + texture, 0, 255))
    return Image.fromarray(np.stack(planes, -1).astype(np.uint8), "RGB")


def run(n=64, w=1024, h=768, quality=90, subsampling=2):
    blobs = []
    for i in range(n):
        buf = io.BytesIO()
        picture(w, h, i).save(buf, "JPEG", quality=quality, subsampling=subsampling)
        blobs.append(buf.getvalue())
    dev = torch.device("cuda:0")
    out = jpeg.decode_many(blobs, dev)
    torch.cuda.synchronize()
    ref = [np.asarray(Image.open(io.BytesIO(b)).convert("RGB")) for b in blobs[:4]]
    exact = all(np.array_equal(o.cpu().numpy(), r) for o, r in zip(out, ref))

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t_dev = timed(lambda: jpeg.decode_many(blobs, dev), 5)
    t_one = timed(lambda: jpeg.decode_many(blobs[:1], dev), 20)
    t_seq = timed(lambda: jpeg.decode_many(blobs[:8], dev, sequential=True), 2)
    t0 = time.perf_counter()
    for b in blobs:
        np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
    t_pil = time.perf_counter() - t0
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    t_ing = timed(lambda: jpeg.ingest_files(blobs, 362, mean, std, device=dev), 3)
    t0 = time.perf_counter()
    for b in blobs:
        img = Image.open(io.BytesIO(b)).convert("RGB")
        img.thumbnail((362, 362), Image.LANCZOS)
        (np.asarray(img).astype(np.float32) / 255.0 - np.array(mean, np.float32)) / np.array(std, np.float32)
    t_pil_ing = time.perf_counter() - t0
    return {"files": n, "size": [w, h], "quality": quality, "subsampling": ["4:4:4", "4:2:2", "4:2:0"][subsampling],
            "mean_file_kb": round(sum(map(len, blobs)) / n / 1024, 1), "byte_exact_vs_pillow": bool(exact),
            "device_decode_files_per_s": round(n / t_dev, 1), "device_decode_megapixels_per_s": round(n * w * h / t_dev / 1e6, 1),
            "device_single_file_ms": round(t_one * 1e3, 3), "one_thread_per_interval_decoder_files_per_s": round(8 / t_seq, 1),
            "pillow_host_1_thread_files_per_s": round(n / t_pil, 1), "speedup_vs_1_host_thread": round(t_pil / t_dev, 1),
            "files_to_normalised_362_tensor_per_s": round(n / t_ing, 1), "pillow_pipeline_1_thread_files_per_s": round(n / t_pil_ing, 1),
            "note": "device figures include header parsing + un-stuffing into the pinned staging buffer (host, one library call each, up to 8 threads) and the upload"}


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    print(json.dumps(run(*a)))
