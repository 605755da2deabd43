"""CLAHE on the GPU (gandtr_amd/csrc/clahe.hip through the C ABI) against oracle/clahe_oracle.py.
The 8-bit core is integer / unfused-float32 arithmetic on both sides: bit-exact.  The float colour conversions differ in the last
bits (device exp2 / log2 vs numpy), which moves an occasional pixel across an 8-bit lightness boundary; the Lab test therefore bounds
the fraction of pixels that differ by more than rounding noise and the size of the largest difference (one lightness level)."""
import numpy as np
import pytest
import torch

from gandtr_amd import clahe
from gandtr_amd.components.data import wrapper as W
from oracle import clahe_oracle as C

pytestmark = pytest.mark.gpu


def _plane(seed, n, h, w, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    if kind == "narrow":                                   # low contrast: most bins empty, heavy clipping
        return np.clip(rng.normal(120, 6, (n, h, w)), 0, 255).astype(np.uint8)
    if kind == "constant":
        return np.full((n, h, w), 201, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)         # smooth ramp + noise, different per image
    base = 127 + 100 * np.sin(xx / w * 5 + seed) * np.cos(yy / h * 3)
    return np.clip(base[None] + rng.normal(0, 8, (n, h, w)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("n,h,w,grid,clip,kind", [
    (2, 64, 64, 8, 1.0, "smooth"), (1, 100, 130, 8, 1.0, "smooth"), (2, 96, 130, 8, 4.0, "narrow"), (3, 256, 256, 8, 1.0, "narrow"),
    (1, 256, 256, 8, 40.0, "uniform"), (1, 61, 67, 4, 2.0, "uniform"), (1, 128, 128, 8, 0.0, "smooth"), (2, 32, 40, 8, 3.0, "constant"),
    (1, 362, 362, 8, 1.0, "smooth"), (2, 1024, 1024, 8, 1.0, "smooth")])
def test_clahe_u8_bit_exact(cuda_device, n, h, w, grid, clip, kind):
    src = _plane(n * 1000 + h + w, n, h, w, kind)
    got = clahe.clahe_u8(torch.from_numpy(src).to(cuda_device), clip, grid).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], C.clahe_u8(src[i], clip, grid, grid)), (i, kind)
    one = clahe.clahe_u8(torch.from_numpy(src[0]).to(cuda_device), clip, grid).cpu().numpy()     # 2-D input, batch independence
    assert np.array_equal(one, got[0])


def _images(seed, n, h, w):
    """generator-like outputs in tanh range: smooth colour fields + texture"""
    g = torch.Generator().manual_seed(seed)
    low = torch.nn.functional.interpolate(torch.randn(n, 3, 9, 9, generator=g), size=(h, w), mode="bicubic", align_corners=False)
    return torch.tanh(0.9 * low + 0.15 * torch.randn(n, 3, h, w, generator=g)).contiguous()


@pytest.mark.parametrize("n,h,w,clip", [(3, 256, 256, 1.0), (2, 100, 130, 4.0), (1, 181, 256, 1.0)])
def test_clahe_lab_matches_oracle(cuda_device, n, h, w, clip):
    x = _images(7 + h, n, h, w)
    mean, std = [0.5, 0.5, 0.5], [0.5, 0.5, 0.5]
    ref = C.clahe_post(x.numpy(), mean, std, clip)
    got = clahe.clahe_lab(x.to(cuda_device), clip, 8, (mean, std), (mean, std)).cpu().numpy()
    diff = np.abs(got - ref)
    assert np.isfinite(got).all()
    assert float((diff > 5e-4).mean()) < 2e-3, float((diff > 5e-4).mean())      # pixels whose 8-bit lightness flipped (or whose LUT did)
    assert float(diff.max()) < 0.05                                              # ... by one level: 100 / 255 in L
    assert float(np.median(diff)) < 2e-5
    assert not np.allclose(got, x.numpy(), atol=1e-2)                            # it did change the image


def test_clahe_post_wrapper_on_device(cuda_device):
    """ClahePost (wrapper.py:325-348) on device tensors: whole batch == image by image, 3-D input, lists, determinism; the augment ->
    clahepost -> meanstd_post order of finetune.yml:13"""
    post = W.WRAPPERS_LABELS["clahepost"]("[[0.5,0.5,0.5],[0.5,0.5,0.5]]", 1.0, device=cuda_device)
    x = _images(11, 4, 256, 256).to(cuda_device)
    y = post.postprocess(x, None, None)
    assert y.shape == x.shape and y.is_cuda
    assert torch.equal(y, post.postprocess(x, None, None))
    for i in range(4):
        assert torch.equal(post.postprocess(x[i], None, None), y[i])
    ys = post.postprocess([x[:2], x[2:]], None, None)
    assert torch.equal(torch.cat(ys), y)
    ref = C.clahe_post(x.cpu().numpy(), [0.5] * 3, [0.5] * 3, 1.0)
    assert float((np.abs(y.cpu().numpy() - ref) > 5e-4).mean()) < 2e-3
    assert post.postprocess(None, None, None) is None
    with pytest.raises(ValueError):
        clahe.clahe_lab(x[:, :2], 1.0)
    with pytest.raises(ValueError):
        clahe.clahe_u8(torch.zeros(4, 4, dtype=torch.uint8, device=cuda_device), 1.0, 8)      # grid larger than the plane


def test_clahe_full_batch_properties(cuda_device):
    """BASELINE config-5 geometry (128 x 3 x 256 x 256): image independence, permutation equivariance, range"""
    x = _images(5, 128, 256, 256).to(cuda_device)
    pair = ([0.5] * 3, [0.5] * 3)
    y = clahe.clahe_lab(x, 1.0, 8, pair, pair)
    assert torch.isfinite(y).all() and float(y.min()) >= -1.0 - 1e-6 and float(y.max()) <= 1.0 + 1e-6
    perm = torch.randperm(128, generator=torch.Generator().manual_seed(0)).to(cuda_device)
    assert torch.equal(clahe.clahe_lab(x[perm], 1.0, 8, pair, pair), y[perm])
    assert torch.equal(clahe.clahe_lab(x[17:18], 1.0, 8, pair, pair), y[17:18])
    unnorm = clahe.clahe_lab(x * 0.5 + 0.5, 1.0, 8)                                            # identity affines
    assert float(((unnorm - 0.5) / 0.5 - y).abs().max()) < 0.05
