"""CPU checks of oracle/clahe_oracle.py.  cv2 is not installed and the reference holds no fixture for this path, so the oracle is
"parity unpinned" (see its header); what can be checked here is that the vectorised restatement agrees with an independent
scalar-loop transcription of the published algorithm, and with published CIE Lab values."""
import math

import numpy as np
import pytest

from oracle import clahe_oracle as C


def _scalar_clahe(src, clip, gx, gy):
    """pixel-by-pixel transcription of cv::CLAHE::apply for 8-bit planes (histogram / clip / redistribute / LUT / blend)"""
    h, w = src.shape
    if w % gx == 0 and h % gy == 0:
        ext = src
    else:
        eh, ew = h + gy - h % gy, w + gx - w % gx
        ext = np.zeros((eh, ew), np.uint8)
        for y in range(eh):
            for x in range(ew):
                ext[y, x] = src[y if y < h else 2 * (h - 1) - y, x if x < w else 2 * (w - 1) - x]
    th, tw = ext.shape[0] // gy, ext.shape[1] // gx
    area = th * tw
    scale = np.float32(255) / np.float32(area)
    limit = max(int(clip * area / 256), 1) if clip > 0 else 0
    luts = [[None] * gx for _ in range(gy)]
    for ty in range(gy):
        for tx in range(gx):
            hist = [0] * 256
            for y in range(ty * th, (ty + 1) * th):
                for x in range(tx * tw, (tx + 1) * tw):
                    hist[ext[y, x]] += 1
            if limit > 0:
                clipped = 0
                for i in range(256):
                    if hist[i] > limit:
                        clipped += hist[i] - limit
                        hist[i] = limit
                batch, residual = clipped // 256, clipped % 256
                hist = [v + batch for v in hist]
                if residual:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1
                        i += step
                        residual -= 1
            lut, s = [], 0
            for i in range(256):
                s += hist[i]
                lut.append(min(max(int(np.rint(np.float32(s) * scale)), 0), 255))
            luts[ty][tx] = lut
    out = np.zeros_like(src)
    inv_tw, inv_th = np.float32(1) / np.float32(tw), np.float32(1) / np.float32(th)
    for y in range(h):
        tyf = np.float32(y) * inv_th - np.float32(0.5)
        ty1 = math.floor(tyf)
        ya = np.float32(tyf - np.float32(ty1)); ya1 = np.float32(1) - ya
        a1, a2 = max(ty1, 0), min(ty1 + 1, gy - 1)
        for x in range(w):
            txf = np.float32(x) * inv_tw - np.float32(0.5)
            tx1 = math.floor(txf)
            xa = np.float32(txf - np.float32(tx1)); xa1 = np.float32(1) - xa
            b1, b2 = max(tx1, 0), min(tx1 + 1, gx - 1)
            v = src[y, x]
            res = (np.float32(luts[a1][b1][v]) * xa1 + np.float32(luts[a1][b2][v]) * xa) * ya1 + \
                  (np.float32(luts[a2][b1][v]) * xa1 + np.float32(luts[a2][b2][v]) * xa) * ya
            out[y, x] = min(max(int(np.rint(res)), 0), 255)
    return out


@pytest.mark.parametrize("h,w,gx,gy,clip", [(32, 32, 4, 4, 2.0), (30, 44, 4, 4, 1.0), (32, 45, 8, 4, 4.0), (17, 16, 2, 3, 0.0),
                                            (24, 24, 8, 8, 40.0)])
def test_vectorised_clahe_equals_scalar_transcription(h, w, gx, gy, clip):
    rng = np.random.default_rng(h * 100 + w)
    src = (np.clip(rng.normal(110, 25, (h, w)), 0, 255)).astype(np.uint8)
    assert np.array_equal(C.clahe_u8(src, clip, gx, gy), _scalar_clahe(src, clip, gx, gy))


def test_clahe_geometry_quirk_of_partial_divisibility():
    assert C.clahe_geometry(64, 64, 8, 8) == (8, 8, 64, 64)
    assert C.clahe_geometry(100, 130, 8, 8) == (13, 17, 104, 136)
    assert C.clahe_geometry(96, 130, 8, 8) == (13, 17, 104, 136)       # the divisible dimension still gains a full `tiles` rows


def test_single_tile_without_clipping_is_histogram_equalisation():
    rng = np.random.default_rng(3)
    src = rng.integers(40, 200, (32, 32)).astype(np.uint8)
    cdf = np.cumsum(np.bincount(src.ravel(), minlength=256)).astype(np.float32)
    want = np.rint(cdf * (np.float32(255) / np.float32(1024))).astype(np.uint8)[src]
    assert np.array_equal(C.clahe_u8(src, 0.0, 1, 1), want)
    luts = C.clahe_luts(src, 2.0, 4, 4)
    assert (np.diff(luts.astype(np.int32), axis=-1) >= 0).all() and (luts[..., -1] == 255).all()   # monotone, full range


def test_constant_plane_maps_to_one_level():
    out = C.clahe_u8(np.full((64, 64), 77, np.uint8), 4.0)
    assert len(np.unique(out)) == 1


def test_lab_conversion_published_values_and_round_trip():
    # CIE L*a*b* (D65) of the sRGB primaries / white / mid grey; OpenCV's documentation lists the same numbers
    rgb = np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [0, 0, 0], [0.5, 0.5, 0.5]]], np.float32)
    want = np.array([[53.2408, 80.0925, 67.2032], [87.7347, -86.1827, 83.1793], [32.2970, 79.1875, -107.8602], [100, 0, 0], [0, 0, 0],
                     [53.3890, 0, 0]], np.float32)
    assert np.abs(C.rgb2lab(rgb)[0] - want).max() < 0.02
    img = np.random.default_rng(0).random((40, 56, 3), dtype=np.float32)
    assert np.abs(C.lab2rgb(C.rgb2lab(img)) - img).max() < 1e-4
    assert np.abs(C.normspace2rgb_lab(C.rgb2normspace_lab(img)) - img).max() < 1e-4
    out_of_range = np.array([[[-0.2, 1.3, 0.5]]], np.float32)                      # inputs are clipped to [0, 1] first
    assert np.allclose(C.rgb2lab(out_of_range), C.rgb2lab(np.clip(out_of_range, 0, 1)))


def test_clahe_post_shapes_and_range():
    x = np.tanh(np.random.default_rng(1).normal(0, 1, (2, 3, 48, 64))).astype(np.float32)
    y = C.clahe_post(x, [0.5] * 3, [0.5] * 3, 1.0)
    assert y.shape == x.shape and y.dtype == np.float32 and np.isfinite(y).all()
    assert y.min() >= -1.0 - 1e-6 and y.max() <= 1.0 + 1e-6
    assert not np.allclose(y[0], y[1])
