"""Golden vectors for the device JPEG decoder: small baseline and progressive files (gray, YCbCr, CMYK / YCCK) written by Pillow and what the reference's loader makes of them,
``np.asarray(Image.open(f).convert('RGB'))`` (pil_loader, mdir/external/cirtorch/datasets/datahelpers.py:39-47).  The decoder the
reference calls IS Pillow (libjpeg-turbo), so these vectors are outputs of the reference's own code path; the tests check the device
decoder against them and, where Pillow is installed, that Pillow still reproduces them (i.e. that the fixture pins the library build).

usage:  python tests/golden/make_jpeg_golden.py            (writes jpeg_cases.npz next to this file)
"""
import io
import os

import numpy as np
import PIL
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))


def picture(w, h, seed, gray=False, cmyk=False):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    planes = []
    for c in range(1 if gray else (4 if cmyk else 3)):
        base = 128 + 90 * np.sin(xx / (5.0 + 3 * c) + seed) * np.cos(yy / (7.0 + c)) + 30 * np.sin((xx + yy) / 23.0)
        noise = rng.normal(0, 25, (h, w)) * (rng.rand(h, w) < 0.4)
        planes.append(np.clip(base + noise + 60.0 * (((xx // 13 + yy // 11) % 2) > 0) - 30, 0, 255))
    arr = np.stack(planes, -1).astype(np.uint8)
    if cmyk:
        return Image.fromarray(arr, "CMYK")
    return Image.fromarray(arr[:, :, 0], "L") if gray else Image.fromarray(arr, "RGB")


def adobe_segment(blob):
    """(offset of the APP14 'Adobe' segment's marker, its total length incl. the marker)"""
    i = 2
    while i + 4 <= len(blob) and blob[i] == 0xFF and blob[i + 1] != 0xDA:
        n = (blob[i + 2] << 8) | blob[i + 3]
        if blob[i + 1] == 0xEE and blob[i + 4:i + 9] == b"Adobe":
            return i, n + 2
        i += 2 + n
    raise ValueError("no Adobe marker")


def as_ycck(blob):
    """the same bitstream declared YCCK: Adobe transform flag 0 -> 2 (the decoder then runs the first three components through its YCbCr -> RGB tables and complements
    them, jdcolor.c ycck_cmyk_convert; Photoshop writes such files, Pillow does not)"""
    i, n = adobe_segment(blob)
    assert blob[i + 15] == 0
    return blob[:i + 15] + b"\x02" + blob[i + 16:]


def without_adobe(blob):
    """the same bitstream without its Adobe marker: four components are then taken as CMYK by the library, and as inverted CMYK by Pillow's plugin all the same"""
    i, n = adobe_segment(blob)
    return blob[:i] + blob[i + n:]


CASES = [  # name, (w, h), gray (True) / four-component kind, save options
    ("c444_q90", (45, 37), False, dict(quality=90, subsampling=0)),
    ("c422_q75", (51, 30), False, dict(quality=75, subsampling=1)),
    ("c420_q85", (64, 48), False, dict(quality=85, subsampling=2)),
    ("c420_q30_odd", (33, 17), False, dict(quality=30, subsampling=2)),
    ("c420_tiny", (3, 2), False, dict(quality=80, subsampling=2)),
    ("c422_narrow", (4, 40), False, dict(quality=80, subsampling=1)),
    ("gray_q70", (40, 29), True, dict(quality=70)),
    ("c420_optimised", (57, 43), False, dict(quality=60, subsampling=2, optimize=True)),
    ("c444_q100", (24, 24), False, dict(quality=100, subsampling=0)),
    ("c420_restart_rows", (70, 50), False, dict(quality=85, subsampling=2, restart_marker_rows=1)),
    ("c444_restart_blocks", (40, 40), False, dict(quality=85, subsampling=0, restart_marker_blocks=3)),
    # progressive (SOF2): spectral selection + successive approximation, libjpeg's default scan script
    ("p420_q85", (64, 48), False, dict(quality=85, subsampling=2, progressive=True)),
    ("p444_q60_odd", (37, 29), False, dict(quality=60, subsampling=0, progressive=True, optimize=True)),
    ("pgray_q75", (40, 29), True, dict(quality=75, progressive=True)),
    ("p422_restart", (51, 30), False, dict(quality=80, subsampling=1, progressive=True, restart_marker_blocks=2)),
    # four components: CMYK as Pillow writes it (Adobe marker, transform 0, inverted samples; with `subsampling` only the first component keeps full resolution),
    # the same declared YCCK, without the Adobe marker, progressive
    ("cmyk444_q90", (45, 37), "cmyk", dict(quality=90)),
    ("cmyk_sub2_q80", (50, 34), "cmyk", dict(quality=80, subsampling=2)),
    ("cmyk_sub1_restart", (41, 30), "cmyk", dict(quality=85, subsampling=1, restart_marker_blocks=4)),
    ("ycck444_q85", (40, 33), "ycck", dict(quality=85)),
    ("ycck_sub2_q75", (64, 48), "ycck", dict(quality=75, subsampling=2)),
    ("cmyk_no_adobe", (36, 28), "noadobe", dict(quality=88)),
    ("pcmyk_q80", (48, 40), "cmyk", dict(quality=80, progressive=True)),
    ("pycck_sub2", (52, 36), "ycck", dict(quality=82, subsampling=2, progressive=True)),
    # progressive four-component files WITH restart intervals (round 5, ADVICE r4: all four DC predictors are reset at a restart)
    ("pcmyk_restart", (44, 36), "cmyk", dict(quality=84, progressive=True, restart_marker_blocks=3)),
    ("pycck_sub2_restart", (52, 36), "ycck", dict(quality=82, subsampling=2, progressive=True, restart_marker_blocks=2)),
]


def main():
    out = {"pillow_version": np.array(PIL.__version__), "libjpeg_version": np.array(str(features.version("jpg"))),
           "libjpeg_turbo": np.array(bool(features.check_feature("libjpeg_turbo"))), "names": np.array([c[0] for c in CASES])}
    for k, (name, (w, h), gray, opts) in enumerate(CASES):
        buf = io.BytesIO()
        picture(w, h, 40 + k, gray is True, isinstance(gray, str)).save(buf, "JPEG", **opts)
        blob = buf.getvalue()
        if gray == "ycck":
            blob = as_ycck(blob)
        elif gray == "noadobe":
            blob = without_adobe(blob)
        with Image.open(io.BytesIO(blob)) as img:
            rgb = np.asarray(img.convert("RGB")).copy()
        out["file_" + name] = np.frombuffer(blob, np.uint8)
        out["rgb_" + name] = rgb
    np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **out)
    print("wrote", len(CASES), "cases,", os.path.getsize(os.path.join(HERE, "jpeg_cases.npz")), "bytes")


if __name__ == "__main__":
    main()
