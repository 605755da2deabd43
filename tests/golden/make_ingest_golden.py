"""Generates tests/golden/ingest.npz with Pillow itself (the third-party library the reference's ingest path calls:
datahelpers.py:75-82 ``img.thumbnail((s, s), LANCZOS)``).  Inputs are small synthetic uint8 images; expected outputs are what
Pillow returns.  Run in the build container:  python tests/golden/make_ingest_golden.py"""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [(60, 80, 36), (75, 50, 36), (231, 77, 50), (64, 64, 64), (65, 64, 64), (150, 225, 18), (250, 167, 20), (49, 400, 36)]


def synth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 127 + 90 * np.cos(xx / 3.0 + yy / 9.0), (xx * 7 + yy * 3) % 256], -1)
    return np.clip(base + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, s) in enumerate(CASES):
        a = synth_image(h, w, i)
        p = Image.fromarray(a)
        p.thumbnail((s, s), Image.LANCZOS)
        out["in_%d" % i] = a
        out["imsize_%d" % i] = np.array(s)
        out["out_%d" % i] = np.asarray(p)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ingest.npz"), **out)
    print("wrote", len(CASES), "cases, Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
