"""CPU oracle for the CLAHE post-processing row -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

What the reference computes (paths relative to /root/reference):
  ClahePost.postprocess          mdir/components/data/wrapper.py:325-348   (un-normalise, per-image CLAHE, normalise)
  ImageClahe.apply               mdir/components/data/transform/functional.py:151-158
  apply_lightness_transform      functional.py:81-85
  rgb2normspace / normspace2rgb  functional.py:28-36 / :55-63  ("lab": cv2.cvtColor float32 RGB<->LAB, L / 100, (a, b + 128) / 255)
  ChannelClahe.apply_clahe       functional.py:147-148  ((chan * 255).astype(uint8) -> cv2 CLAHE -> / 255)

**PARITY UNPINNED.**  All of the arithmetic lives in opencv-python (`cv2`), a dependency the reference lists without a
version pin (requirements.txt: `opencv-python`) and which is not installed in the build container; the reference holds no
test, fixture or golden image for this path.  This module restates the *published* algorithms:
  * CLAHE as implemented by OpenCV 4.x `cv::CLAHE::apply` for 8-bit input (modules/imgproc/src/clahe.cpp): tile grid with
    BORDER_REFLECT_101 extension (including the library's quirk of adding a full extra `tiles` columns/rows when only the
    OTHER dimension is indivisible), per-tile 256-bin histogram, integer clip limit `max(int(clip * area / 256), 1)`, excess
    redistributed as `excess // 256` everywhere plus a strided +1 over the first `residual` slots, LUT =
    round-half-even(cumsum * 255 / area), float32 bilinear blend of the four neighbouring tiles' LUTs;
  * float32 `COLOR_RGB2LAB` / `COLOR_LAB2RGB` as the direct CIE formulas with OpenCV's constants (sRGB D65 matrices, white
    point (0.950456, 1, 1.088754), thresholds 0.008856 / 7.787 / 903.3) and the exact sRGB transfer curve.  OpenCV itself
    evaluates the transfer curve through a 1024-knot spline and, in SIMD builds, the whole RGB->Lab map through a trilinear
    table, so cv2's floats differ from these in the low decimal places; after the 8-bit quantisation of L that moves a small
    fraction of pixels by one level.
The integer CLAHE core is exact integer / float32 arithmetic here and in the HIP path, so those two are compared bit for bit.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np

_F = np.float32

# OpenCV color_lab.cpp constants
_RGB2XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float64)
_XYZ2RGB = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float64)
_WHITE = np.array([0.950456, 1.0, 1.088754], np.float64)


def lab_matrices():
    """(forward 3x3, inverse 3x3) in float32: rows of RGB->XYZ divided by the white point; columns of XYZ->RGB multiplied by it."""
    fwd = (_RGB2XYZ / _WHITE[:, None]).astype(_F)
    inv = (_XYZ2RGB * _WHITE[None, :]).astype(_F)
    return fwd, inv


def srgb_to_linear(c):
    c = c.astype(_F)
    hi = np.power((c + _F(0.055)) * _F(1.0 / 1.055), _F(2.4), dtype=_F)
    return np.where(c <= _F(0.04045), c * _F(1.0 / 12.92), hi).astype(_F)


def linear_to_srgb(c):
    c = c.astype(_F)
    hi = np.power(np.maximum(c, _F(0)), _F(1.0 / 2.4), dtype=_F) * _F(1.055) - _F(0.055)
    return np.where(c <= _F(0.0031308), c * _F(12.92), hi).astype(_F)


def rgb2lab(img):
    """cv2.cvtColor(float32 HxWx3 in [0, 1], COLOR_RGB2LAB): L in [0, 100], a / b roughly in [-127, 127]."""
    fwd, _ = lab_matrices()
    lin = srgb_to_linear(np.clip(img.astype(_F), _F(0), _F(1)))
    r, g, b = lin[..., 0], lin[..., 1], lin[..., 2]
    xyz = [fwd[i, 0] * r + fwd[i, 1] * g + fwd[i, 2] * b for i in range(3)]
    f = [np.where(v > _F(0.008856), np.cbrt(v, dtype=_F), _F(7.787) * v + _F(16.0 / 116.0)).astype(_F) for v in xyz]
    L = np.where(xyz[1] > _F(0.008856), _F(116.0) * f[1] - _F(16.0), _F(903.3) * xyz[1]).astype(_F)
    return np.stack([L, _F(500.0) * (f[0] - f[1]), _F(200.0) * (f[1] - f[2])], axis=-1).astype(_F)


def lab2rgb(lab):
    """cv2.cvtColor(float32 Lab, COLOR_LAB2RGB): result clipped to [0, 1] before the transfer curve, as OpenCV does."""
    _, inv = lab_matrices()
    L, a, b = (lab[..., i].astype(_F) for i in range(3))
    l_thresh, f_thresh = _F(0.008856 * 903.3), _F(7.787 * 0.008856 + 16.0 / 116.0)
    y_lo = L / _F(903.3)
    fy_hi = (L + _F(16.0)) / _F(116.0)
    low = L <= l_thresh
    y = np.where(low, y_lo, fy_hi * fy_hi * fy_hi).astype(_F)
    fy = np.where(low, _F(7.787) * y_lo + _F(16.0 / 116.0), fy_hi).astype(_F)
    out = []
    for fv in (a / _F(500.0) + fy, fy - b / _F(200.0)):
        out.append(np.where(fv <= f_thresh, (fv - _F(16.0 / 116.0)) / _F(7.787), fv * fv * fv).astype(_F))
    x, z = out
    rgb = [np.clip(inv[i, 0] * x + inv[i, 1] * y + inv[i, 2] * z, _F(0), _F(1)) for i in range(3)]
    return np.stack([linear_to_srgb(c) for c in rgb], axis=-1).astype(_F)


def rgb2normspace_lab(img):
    """functional.py:28-36, colorspace 'lab'"""
    return ((rgb2lab(img) + np.array([0, 128, 128], _F)) / np.array([100.0, 255.0, 255.0], _F)).astype(_F)


def normspace2rgb_lab(spc):
    """functional.py:55-63, colorspace 'lab'"""
    return lab2rgb(spc * np.array([100.0, 255.0, 255.0], _F) - np.array([0, 128, 128], _F))


def quantise_u8(chan):
    """(chan * 255).astype(np.uint8), functional.py:148 (truncation; chan is in [0, 1])"""
    return (chan.astype(_F) * _F(255)).astype(np.uint8)


def _reflect101(idx, size):
    return np.where(idx < size, idx, 2 * (size - 1) - idx)


def clahe_geometry(h, w, tiles_x, tiles_y):
    """tile size of cv::CLAHE::apply: the image is extended (REFLECT_101) unless BOTH dimensions divide."""
    if w % tiles_x == 0 and h % tiles_y == 0:
        return h // tiles_y, w // tiles_x, h, w
    eh, ew = h + tiles_y - h % tiles_y, w + tiles_x - w % tiles_x
    return eh // tiles_y, ew // tiles_x, eh, ew


def clahe_luts(src, clip_limit, tiles_x, tiles_y):
    """per-tile lookup tables [tiles_y][tiles_x][256] uint8 of cv::CLAHE (CLAHE_CalcLut_Body)"""
    h, w = src.shape
    th, tw, eh, ew = clahe_geometry(h, w, tiles_x, tiles_y)
    ext = src[_reflect101(np.arange(eh), h)][:, _reflect101(np.arange(ew), w)]
    area = th * tw
    lut_scale = _F(255) / _F(area)
    limit = 0
    if clip_limit > 0.0:
        limit = max(int(float(clip_limit) * area / 256), 1)
    luts = np.empty((tiles_y, tiles_x, 256), np.uint8)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if limit > 0:
                clipped = int(np.maximum(hist - limit, 0).sum())
                hist = np.minimum(hist, limit)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1
                        i += step
                        residual -= 1
            csum = np.cumsum(hist).astype(_F)
            luts[ty, tx] = np.clip(np.rint(csum * lut_scale), 0, 255).astype(np.uint8)     # saturate_cast<uchar>(float): cvRound
    return luts


def clahe_u8(src, clip_limit, tiles_x=8, tiles_y=8):
    """cv2.createCLAHE(clipLimit, (tiles_x, tiles_y)).apply(src) for a uint8 HxW plane (CLAHE_Interpolation_Body)."""
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    th, tw, _, _ = clahe_geometry(h, w, tiles_x, tiles_y)
    luts = clahe_luts(src, clip_limit, tiles_x, tiles_y).astype(_F)

    def axis(n, t, tiles):
        f = np.arange(n, dtype=_F) * (_F(1.0) / _F(t)) - _F(0.5)
        i1 = np.floor(f).astype(np.int64)
        a = (f - i1.astype(_F)).astype(_F)
        return np.maximum(i1, 0), np.minimum(i1 + 1, tiles - 1), a, (_F(1.0) - a).astype(_F)

    ty1, ty2, ya, ya1 = axis(h, th, tiles_y)
    tx1, tx2, xa, xa1 = axis(w, tw, tiles_x)
    v = src.astype(np.int64)
    Y1, Y2, X1, X2 = ty1[:, None], ty2[:, None], tx1[None, :], tx2[None, :]
    XA, XA1, YA, YA1 = xa[None, :], xa1[None, :], ya[:, None], ya1[:, None]
    res = (luts[Y1, X1, v] * XA1 + luts[Y1, X2, v] * XA) * YA1 + (luts[Y2, X1, v] * XA1 + luts[Y2, X2, v] * XA) * YA
    assert res.dtype == _F
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def image_clahe(img, clip_limit, grid_size=8):
    """ImageClahe(clip_limit, grid_size, 'lab').apply(img) for a float32 HxWx3 RGB image (functional.py:81-85,147-158)"""
    spc = rgb2normspace_lab(img)
    spc[..., 0] = clahe_u8(quantise_u8(spc[..., 0]), clip_limit, grid_size, grid_size).astype(_F) / _F(255.0)
    return normspace2rgb_lab(spc)


def clahe_post(x, mean, std, clip_limit, grid_size=8):
    """ClahePost.postprocess on an N x 3 x H x W float32 array (wrapper.py:334-348): per image x * std + mean -> CLAHE ->
    (. - mean) / std."""
    mean = np.asarray(mean, _F).reshape(3, 1, 1)
    std = np.asarray(std, _F).reshape(3, 1, 1)
    out = np.empty_like(x, dtype=_F)
    for i in range(x.shape[0]):
        img = (x[i].astype(_F) * std + mean).transpose(1, 2, 0)
        out[i] = (image_clahe(img, clip_limit, grid_size).transpose(2, 0, 1) - mean) / std
    return out
