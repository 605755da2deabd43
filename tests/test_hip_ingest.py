"""Image ingest on the GPU (gandtr_amd/csrc/ingest.hip through the C ABI): bit-exact against Pillow's golden vectors, against the
oracle (itself pinned to Pillow) and -- Pillow being installed on the GPU box -- against a live Pillow call."""
import os

import numpy as np
import pytest
import torch

from gandtr_amd import ingest
from oracle import ingest_oracle as I

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ingest.npz")
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def test_golden_vectors_from_pillow(cuda_device):
    g = np.load(GOLD)
    n = sum(1 for k in g.files if k.startswith("in_"))
    for i in range(n):
        got = ingest.imresize(torch.from_numpy(g["in_%d" % i]).to(cuda_device), int(g["imsize_%d" % i])).cpu().numpy()
        assert got.shape == g["out_%d" % i].shape and np.array_equal(got, g["out_%d" % i]), i


@pytest.mark.parametrize("h,w,s", [(600, 800, 362), (333, 500, 362), (500, 375, 362), (1200, 1600, 1024), (1025, 1024, 1024),
                                   (2500, 1667, 362), (2001, 2999, 362), (97, 4000, 362), (1500, 200, 128), (100, 37, 50), (64, 64, 64),
                                   (3000, 4000, 1024), (768, 1024, 1024)])
def test_imresize_bit_exact(cuda_device, h, w, s):
    a = np.random.default_rng(h + w).integers(0, 256, (h, w, 3)).astype(np.uint8)
    ref = I.imresize(a, s)
    got = ingest.imresize(torch.from_numpy(a).to(cuda_device), s).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)
    try:
        from PIL import Image
    except ImportError:
        return
    p = Image.fromarray(a)
    p.thumbnail((s, s), Image.LANCZOS)
    assert np.array_equal(got, np.asarray(p))


@pytest.mark.parametrize("h,w,ow,oh", [(60, 80, 200, 150), (100, 100, 100, 37), (33, 47, 47, 33), (64, 48, 48, 200)])
def test_plain_resize_including_enlargement(cuda_device, h, w, ow, oh):
    """Image.resize((ow, oh), LANCZOS) without the thumbnail plan: enlargements use the unit-support filter (filterscale = 1)"""
    a = np.random.default_rng(h * w).integers(0, 256, (h, w, 3)).astype(np.uint8)
    got, _ = ingest.resize(torch.from_numpy(a).to(cuda_device), ow, oh)
    assert np.array_equal(got.cpu().numpy(), I.resample_u8(a, ow, oh))
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(got.cpu().numpy(), np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS)))


@pytest.mark.parametrize("c", [1, 2, 4])
def test_other_channel_counts(cuda_device, c):
    a = np.random.default_rng(c).integers(0, 256, (300, 411, c)).astype(np.uint8)
    ref = I.resample_u8(I.reduce_u8(a, 2, 3), 90, 41, (0.0, 0.0, 411 / 2, 100.0))
    got, _ = ingest.resize(torch.from_numpy(a).to(cuda_device), 90, 41, (2, 3), (0.0, 0.0, 411 / 2, 100.0))
    assert np.array_equal(got.cpu().numpy(), ref)


def test_fused_tensor_output_is_exact(cuda_device):
    """imresize + pil2np | totensor | normalize (core_transforms.py:35-100): fp32 CHW, bit-identical to the numpy / torch CPU ops"""
    a = np.random.default_rng(5).integers(0, 256, (700, 933, 3)).astype(np.uint8)
    ref = I.to_tensor_normalize(I.imresize(a, 362), MEAN, STD)
    got = ingest.ingest(torch.from_numpy(a).to(cuda_device), 362, MEAN, STD).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32 and np.array_equal(got, ref)
    small = np.random.default_rng(6).integers(0, 256, (120, 100, 3)).astype(np.uint8)            # already fits: conversions only
    got = ingest.ingest(torch.from_numpy(small).to(cuda_device), 362, MEAN, STD).cpu().numpy()
    assert np.array_equal(got, I.to_tensor_normalize(small, MEAN, STD))
    assert ingest.imresize(torch.from_numpy(small).to(cuda_device), 362).shape == (120, 100, 3)


def test_hub_transform_chain_with_clahe(cuda_device):
    """hub transform `pil2np | apply_clahe:1.0 | totensor | normalize` (embedding.yml:14) after imresize, on the device"""
    from oracle import clahe_oracle as C
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
    a = np.stack([127 + 100 * np.sin(xx / 90) * np.cos(yy / 70), 127 + 90 * np.cos(xx / 50 + yy / 110), 60 + 0.25 * xx], -1)
    a = np.clip(a + np.random.default_rng(1).normal(0, 6, a.shape), 0, 255).astype(np.uint8)
    small = I.imresize(a, 256)
    ref = ((C.image_clahe(small.astype(np.float32) / np.float32(255), 1.0, 8).transpose(2, 0, 1)
            - np.asarray(MEAN, np.float32)[:, None, None]) / np.asarray(STD, np.float32)[:, None, None])
    got = ingest.ingest(torch.from_numpy(a).to(cuda_device), 256, MEAN, STD, clahe_clip=1.0).cpu().numpy()
    diff = np.abs(got - ref)
    assert got.shape == ref.shape
    assert float((diff > 2e-3).mean()) < 2e-3 and float(diff.max()) < 0.2      # CLAHE tolerance of tests/test_hip_clahe.py, in 1/std units


def test_ingest_many_matches_one_by_one(cuda_device):
    """a list of mixed sizes through the batched call == the same images one by one (bitwise), in input order"""
    rng = np.random.default_rng(9)
    shapes = [(600, 800), (333, 500), (1200, 1600), (500, 375), (768, 1024), (2001, 1333), (97, 400), (640, 640), (1024, 683)]
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).to(cuda_device) for h, w in shapes]
    one = [ingest.ingest(im, 362, MEAN, STD) for im in imgs]
    many = ingest.ingest_many(imgs, 362, MEAN, STD, streams=4)
    assert len(many) == len(one)
    for a, b in zip(one, many):
        assert a.shape == b.shape and torch.equal(a, b)
    again = ingest.ingest_many(imgs, 362, MEAN, STD, clahe_clip=1.0, streams=3)
    ref = [ingest.ingest(im, 362, MEAN, STD, clahe_clip=1.0) for im in imgs]
    for a, b in zip(ref, again):
        assert torch.equal(a, b)
    assert ingest.ingest_many([], 362, MEAN, STD) == []


def test_resize_many_matches_resize(cuda_device):
    """the batched entry point (one call, three launches for RGB lists) against the per-image one, bitwise: plain thumbnails, a
    box reduction (3000 x 2000 -> 362), an image that needs no resampling, a cropped box, uint8 + fp32 outputs; and a grayscale
    list, which runs image by image inside the call"""
    rng = np.random.default_rng(10)
    shapes = [(600, 800), (2000, 3000), (362, 362), (97, 400), (1024, 683), (500, 375)]
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).to(cuda_device) for h, w in shapes]
    plans = [ingest._plan(im, 362) for im in imgs]
    assert any(p[1] != (1, 1) for p in plans) and any(p[0] == (im.shape[1], im.shape[0]) for p, im in zip(plans, imgs))
    plans[0] = ((300, 200), (1, 1), (10.0, 20.0, 700.0, 500.0))                      # explicit crop box + its own output size
    u8, chw = ingest.resize_many(imgs, plans, want_u8=True, mean_std=(MEAN, STD), want_chw=True)
    for im, (size, factors, box), a, b in zip(imgs, plans, u8, chw):
        ra, rb = ingest.resize(im, size[0], size[1], factors, box, want_u8=True, mean_std=(MEAN, STD), want_chw=True)
        assert a.shape == ra.shape and torch.equal(a, ra)
        assert b.shape == rb.shape and torch.equal(b, rb)
    gray = [torch.from_numpy(rng.integers(0, 256, (h, w, 1)).astype(np.uint8)).to(cuda_device) for h, w in shapes[:3]]
    gplans = [ingest._plan(im, 200) for im in gray]
    gu8, _ = ingest.resize_many(gray, gplans)
    for im, (size, factors, box), a in zip(gray, gplans, gu8):
        assert torch.equal(a, ingest.resize(im, size[0], size[1], factors, box)[0])
    with pytest.raises(ValueError):
        ingest.resize_many([imgs[0], gray[0]], [plans[0], gplans[0]])                 # mixed channel counts


def test_resize_many_across_the_coefficient_cache_limit(cuda_device):
    """The device-resident coefficient tables are evicted only between calls: a batched call with more distinct geometries than the
    cache holds (256 tables; 150 images of distinct sizes need up to 300) must still hand every item live tables -- bitwise equal
    to the per-image call, which is evaluated AFTER the batch so that the batch ran on a cache pre-filled by other geometries."""
    rng = np.random.default_rng(11)
    warm = [torch.from_numpy(rng.integers(0, 256, (40 + i, 53 + 2 * i, 3)).astype(np.uint8)).to(cuda_device) for i in range(120)]
    for im in warm:                                                                   # ~240 tables of other geometries
        ingest.imresize(im, 32)
    shapes = [(64 + 3 * i, 200 - i) for i in range(150)]
    imgs = [torch.from_numpy(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).to(cuda_device) for h, w in shapes]
    plans = [ingest._plan(im, 48) for im in imgs]
    u8, _ = ingest.resize_many(imgs, plans)
    torch.cuda.synchronize()
    for im, (size, factors, box), a in zip(imgs, plans, u8):
        assert torch.equal(a, ingest.resize(im, size[0], size[1], factors, box)[0])


def test_hub_device_transform(cuda_device):
    """net.transform_device(decoded uint8 image) == ingest with the network's own mean / std / CLAHE settings, and feeds the net"""
    import hubconf
    net = hubconf.gem_vgg16_cyclegan(pretrained=False, device=cuda_device)
    a = np.random.default_rng(4).integers(0, 256, (300, 400, 3)).astype(np.uint8)
    u8 = torch.from_numpy(a).to(cuda_device)
    x = net.transform_device(u8, imsize=256)
    assert x.shape == (3, 192, 256) and x.is_cuda
    assert torch.equal(x, ingest.ingest(u8, 256, MEAN, STD, clahe_clip=1.0))
    with torch.no_grad():
        d = net(x[None])
    assert d.shape[0] == 512 and torch.isfinite(d).all()


def test_argument_errors(cuda_device):
    with pytest.raises(ValueError):
        ingest.imresize(torch.zeros(8, 8, 3), 4)                                        # not on the device
    with pytest.raises(ValueError):
        ingest.resize(torch.zeros(8, 8, 5, dtype=torch.uint8, device=cuda_device), 4, 4)   # > 4 channels
    with pytest.raises(ValueError):
        ingest.resize(torch.zeros(8, 8, 3, dtype=torch.uint8, device=cuda_device), 4, 4, (16, 1))   # factor larger than the image
