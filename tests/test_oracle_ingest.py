"""oracle/ingest_oracle.py against Pillow (the library the reference's ingest path calls): the committed golden vectors, and --
when Pillow is importable, as it is in the build image and on the GPU box -- a live sweep of sizes incl. the reducing-gap path."""
import os

import numpy as np
import pytest

from oracle import ingest_oracle as I

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ingest.npz")


def test_golden_vectors_from_pillow():
    g = np.load(GOLD)
    n = sum(1 for k in g.files if k.startswith("in_"))
    assert n >= 8
    for i in range(n):
        got = I.imresize(g["in_%d" % i], int(g["imsize_%d" % i]))
        assert got.shape == g["out_%d" % i].shape and np.array_equal(got, g["out_%d" % i]), i


@pytest.mark.parametrize("h,w,s", [(600, 800, 362), (333, 500, 362), (500, 375, 362), (1200, 1600, 1024), (1025, 1024, 1024),
                                   (2500, 1667, 362), (2001, 2999, 362), (97, 4000, 362), (1500, 200, 128), (100, 37, 50), (64, 64, 64)])
def test_imresize_equals_pillow_thumbnail(h, w, s):
    Image = pytest.importorskip("PIL.Image")
    a = np.random.default_rng(h + w).integers(0, 256, (h, w, 3)).astype(np.uint8)
    p = Image.fromarray(a)
    p.thumbnail((s, s), Image.LANCZOS)
    got = I.imresize(a, s)
    assert got.shape == np.asarray(p).shape and np.array_equal(got, np.asarray(p))


@pytest.mark.parametrize("f", [(2, 2), (3, 3), (5, 5), (3, 2), (1, 3), (7, 3)])
def test_reduce_equals_pillow(f):
    Image = pytest.importorskip("PIL.Image")
    for (h, w) in [(30, 30), (31, 30), (29, 34)]:
        a = np.random.default_rng(h * w).integers(0, 256, (h, w, 3)).astype(np.uint8)
        assert np.array_equal(I.reduce_u8(a, f[0], f[1]), np.asarray(Image.fromarray(a).reduce(f)))


def test_thumbnail_size_rule_and_plan():
    assert I.thumbnail_size(800, 600, 362) == (362, 272)
    assert I.thumbnail_size(375, 500, 362) == (271, 362)
    assert I.thumbnail_size(300, 200, 362) is None                      # never enlarges
    assert I.reduce_plan(3000, 2000, 362, 241) == (4, 4, (0.0, 0.0, 750.0, 500.0))
    assert I.reduce_plan(800, 600, 362, 272)[:2] == (1, 1)
    from gandtr_amd import ingest                                        # the product's host-side plan is the same rule
    for (w, h, s) in [(800, 600, 362), (375, 500, 362), (3000, 2000, 362), (4001, 2999, 1024), (300, 200, 362)]:
        assert ingest.thumbnail_size(w, h, s) == I.thumbnail_size(w, h, s)
        size = I.thumbnail_size(w, h, s)
        if size:
            assert ingest.reduce_plan(w, h, *size) == I.reduce_plan(w, h, *size)


def test_to_tensor_normalize_matches_torch_ops():
    import torch
    a = np.random.default_rng(0).integers(0, 256, (7, 9, 3)).astype(np.uint8)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    t = torch.from_numpy(a.astype(np.float32) / 255.0).permute(2, 0, 1)          # Pil2Numpy + ToTensor
    want = (t - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
    assert np.array_equal(I.to_tensor_normalize(a, mean, std), want.numpy())
