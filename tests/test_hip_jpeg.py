"""Device JPEG decoder (gandtr_amd/csrc/jpeg.hip through the C ABI) against the reference's loader, pil_loader =
Image.open(f).convert('RGB') (mdir/external/cirtorch/datasets/datahelpers.py:39-47): byte-exact, every sampling mode Pillow writes,
odd sizes, custom Huffman tables, restart intervals, both entropy decoders (parallel pieces / one thread per interval)."""
import io
import os

import numpy as np
import pytest
import torch

from gandtr_amd import jpeg

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")


def _picture(w, h, seed, gray=False):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    planes = []
    for c in range(1 if gray else 3):
        base = 128 + 90 * np.sin(xx / (5.0 + 3 * c) + seed) * np.cos(yy / (7.0 + c)) + 30 * np.sin((xx + yy) / 23.0)
        noise = rng.normal(0, 25, (h, w)) * (rng.rand(h, w) < 0.4)
        edges = 80.0 * (((xx // 13 + yy // 11) % 2) > 0)
        planes.append(np.clip(base + noise + edges - 40, 0, 255))
    arr = np.stack(planes, -1).astype(np.uint8)
    return Image.fromarray(arr[:, :, 0], "L") if gray else Image.fromarray(arr, "RGB")


def _encode(img, **kw):
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 22)      # (Pillow's progressive / optimising encoder needs the whole file in one buffer)
    buf = io.BytesIO()
    img.save(buf, "JPEG", **kw)
    return buf.getvalue()


def _reference(blob):
    with Image.open(io.BytesIO(blob)) as img:
        return np.asarray(img.convert("RGB")).copy()


def _check(blobs, sequential=False):
    got = jpeg.decode_many(blobs, "cuda:0", sequential=sequential)
    torch.cuda.synchronize()
    for k, (g, blob) in enumerate(zip(got, blobs)):
        want = _reference(blob)
        g = g.cpu().numpy()
        assert g.shape == want.shape, (k, g.shape, want.shape)
        bad = np.argwhere(g != want)
        assert bad.size == 0, "image %d (%dx%d): %d differing samples, first at %s: %s vs %s" % (
            k, want.shape[1], want.shape[0], len(bad), bad[0], g[tuple(bad[0])], want[tuple(bad[0])])


SIZES = [(1, 1), (2, 3), (7, 5), (8, 8), (16, 16), (17, 33), (5, 64), (64, 48), (131, 257), (320, 213)]


@pytest.mark.parametrize("sequential", [True, False], ids=["one-thread-per-interval", "parallel-pieces"])
@pytest.mark.parametrize("subsampling", [0, 1, 2], ids=["444", "422", "420"])
def test_colour_files_match_pillow(subsampling, sequential):
    blobs = [_encode(_picture(w, h, 3 * i + subsampling), quality=q, subsampling=subsampling)
             for i, (w, h) in enumerate(SIZES) for q in (35, 90)]
    _check(blobs, sequential)


@pytest.mark.parametrize("sequential", [True, False], ids=["one-thread-per-interval", "parallel-pieces"])
def test_grayscale_optimised_tables_and_extreme_qualities(sequential):
    blobs = [_encode(_picture(w, h, 7 + i, gray=True), quality=80) for i, (w, h) in enumerate(SIZES)]
    blobs += [_encode(_picture(97, 61, 5), quality=q, subsampling=s, optimize=True) for q in (5, 50, 100) for s in (0, 2)]
    blobs += [_encode(_picture(40, 40, 9), quality=100, subsampling=0), _encode(_picture(333, 100, 11), quality=1, subsampling=2)]
    _check(blobs, sequential)


@pytest.mark.parametrize("sequential", [True, False], ids=["one-thread-per-interval", "parallel-pieces"])
def test_restart_intervals(sequential):
    blobs = []
    for i, (kw, sub) in enumerate([({"restart_marker_blocks": 1}, 2), ({"restart_marker_blocks": 7}, 0), ({"restart_marker_rows": 1}, 2),
                                   ({"restart_marker_rows": 2}, 1), ({"restart_marker_blocks": 1000000}, 2)]):
        blobs.append(_encode(_picture(150 + 9 * i, 90 + 5 * i, 20 + i), quality=85, subsampling=sub, **kw))
    blobs.append(_encode(_picture(64, 64, 31, gray=True), quality=70, restart_marker_blocks=3))
    infos = [jpeg.parse(b).info for b in blobs]
    assert infos[0].nsegments > 1 and infos[2].nsegments > 1 and infos[4].nsegments == 1
    _check(blobs, sequential)


def test_large_noisy_image_many_pieces_converges():
    """a scan of about a megabyte = thousands of 128-byte pieces per file; high-entropy content gives the self-synchronisation
    the least help (few zero runs / end-of-block symbols)"""
    rng = np.random.RandomState(0)
    noisy = Image.fromarray(rng.randint(0, 256, (900, 1400, 3), dtype=np.uint8), "RGB")
    photo = _picture(1600, 1200, 2)
    blobs = [_encode(noisy, quality=95, subsampling=0), _encode(photo, quality=92, subsampling=2), _encode(noisy, quality=60, subsampling=2)]
    assert len(blobs[0]) > 1000000
    _check(blobs, False)


def test_batch_of_mixed_files_equals_one_by_one():
    blobs = [_encode(_picture(200 + 17 * i, 100 + 31 * i, i), quality=70 + 3 * i, subsampling=i % 3) for i in range(7)]
    blobs.insert(3, _encode(_picture(77, 191, 99, gray=True), quality=88))
    together = jpeg.decode_many(blobs, "cuda:0")
    for t, b in zip(together, blobs):
        assert torch.equal(t, jpeg.decode(b, "cuda:0"))
    _check(blobs)


def test_unsupported_files_raise_and_host_loader_is_explicit():
    cmyk = _encode(_picture(32, 32, 2), quality=80, keep_rgb=True)        # (an RGB-coded file: no colour transform -- one of the kinds the device decoder leaves to the host)
    png = io.BytesIO()
    _picture(20, 20, 3).save(png, "PNG")
    for blob, word in ((cmyk, "RGB-coded"), (png.getvalue(), "SOI")):
        with pytest.raises(ValueError, match=word):
            jpeg.parse(blob)
    good = _encode(_picture(48, 40, 4), quality=90)
    with pytest.raises(ValueError):
        jpeg.load_many([good, cmyk], "cuda:0")
    calls = []

    def loader(data):
        calls.append(len(data))
        return _reference(data)

    out = jpeg.load_many([good, cmyk], "cuda:0", host_loader=loader)
    assert calls == [len(cmyk)]
    assert np.array_equal(out[0].cpu().numpy(), _reference(good)) and np.array_equal(out[1].cpu().numpy(), _reference(cmyk))


@pytest.mark.parametrize("subsampling", [0, 1, 2], ids=["444", "422", "420"])
def test_progressive_files_match_pillow(subsampling):
    """SOF2 files (spectral selection + successive approximation: what Pillow writes with progressive=True, libjpeg's default scan script of
    10 scans for colour) -- coefficients by the library's host decoder, everything behind it on the device: byte-exact with pil_loader"""
    blobs = [_encode(_picture(w, h, 70 + k), quality=q, subsampling=subsampling, progressive=True, optimize=(k % 2 == 0))
             for k, ((w, h), q) in enumerate(zip(SIZES, [90, 75, 50, 95, 85, 30, 60, 100, 80, 92]))]
    for b in blobs:
        assert jpeg.parse(b).info.progressive == 1
    _check(blobs)


def test_progressive_grayscale_restarts_and_mixed_lists():
    gray = [_encode(_picture(w, h, 90 + k, gray=True), quality=q, progressive=True) for k, ((w, h), q) in enumerate(zip(SIZES[2:8], [95, 20, 70, 85, 50, 99]))]
    _check(gray)
    # restart intervals inside progressive scans (end-of-band runs and DC predictions restart with them)
    rst = [_encode(_picture(96, 80, 120 + k), quality=85, subsampling=s, progressive=True, restart_marker_blocks=b) for k, (s, b) in enumerate([(0, 3), (2, 2), (1, 5)])]
    _check(rst)
    # one list of baseline and progressive files: one library call per kind, results in input order
    mixed = [_encode(_picture(64, 48, 130), quality=80), rst[0], _encode(_picture(31, 45, 131), quality=90, progressive=True),
             _encode(_picture(80, 64, 132, gray=True), quality=70), gray[1]]
    _check(mixed)
    with pytest.raises(ValueError, match="truncated"):
        jpeg.decode_many([rst[1][: 2 * len(rst[1]) // 3]], "cuda:0")


def test_files_to_network_input_matches_the_pillow_pipeline():
    """decode + imresize + totensor + normalize on the device == the reference's loader + transform on the host
    (genericdataset.py:66-102): the decoded pixels are identical, so the existing ingest parity carries over; checked end to end here"""
    from gandtr_amd import ingest
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    blobs = [_encode(_picture(500, 375, 1), quality=90, subsampling=2), _encode(_picture(300, 420, 2), quality=85, subsampling=0)]
    got = jpeg.ingest_files(blobs, 256, mean, std, device="cuda:0")
    for g, b in zip(got, blobs):
        with Image.open(io.BytesIO(b)) as img:
            img = img.convert("RGB")
            img.thumbnail((256, 256), Image.LANCZOS)
            want = (np.asarray(img).astype(np.float32) / 255.0 - np.array(mean, np.float32)) / np.array(std, np.float32)
        assert np.abs(g.cpu().numpy() - want.transpose(2, 0, 1)).max() < 1e-5


def test_golden_files():
    """the committed vectors (tests/golden/jpeg_cases.npz: files + pil_loader's output, make_jpeg_golden.py), both entropy decoders"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_cases.npz"))
    blobs = [bytes(g["file_" + str(n)]) for n in g["names"]]
    for sequential in (False, True):
        got = jpeg.decode_many(blobs, "cuda:0", sequential=sequential)
        for n, t in zip(g["names"], got):
            assert np.array_equal(t.cpu().numpy(), g["rgb_" + str(n)]), (str(n), sequential)


def test_four_component_files_match_pillow():
    """CMYK files as Pillow writes them (Adobe marker, transform 0; with `subsampling` only the first component keeps full resolution), the same bitstreams declared
    YCCK (transform 2: what Photoshop writes) or stripped of the Adobe marker, baseline and progressive, with restart intervals, in one mixed call with YCbCr and
    gray files: byte-identical to Image.open(f).convert('RGB') (pil_loader, datahelpers.py:39-47)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_jpeg_golden import as_ycck, without_adobe
    rng = np.random.RandomState(11)
    blobs = []
    for i, (w, h) in enumerate([(64, 48), (97, 61), (33, 17), (120, 90), (8, 8), (1, 1)]):
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.stack([(xx * 3 + yy * (c + 1) * 2 + 40 * c) % 256 for c in range(4)], -1).astype(np.float64) + rng.normal(0, 12, (h, w, 4))
        img = Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8), "CMYK")
        for opts in (dict(quality=90), dict(quality=75, subsampling=2), dict(quality=85, subsampling=1, restart_marker_blocks=3), dict(quality=80, progressive=True),
                     dict(quality=70, subsampling=2, progressive=True)):
            blob = _encode(img, **opts)
            blobs += [blob, as_ycck(blob)]
        blobs.append(without_adobe(_encode(img, quality=88)))
    blobs.append(_encode(_picture(50, 40, 9), quality=85, subsampling=2))
    blobs.append(_encode(_picture(40, 30, 10).convert("L"), quality=80))
    for sequential in (False, True):
        got = jpeg.decode_many(blobs, "cuda:0", sequential=sequential)
        for k, (t, blob) in enumerate(zip(got, blobs)):
            assert np.array_equal(t.cpu().numpy(), _reference(blob)), (k, sequential)


def test_corrupted_entropy_data_decodes_to_something_without_harm():
    """damaged scans (random bytes overwritten, none of them 0xFF so the markers stay where they are): the decoder must finish, keep to its
    buffers and leave the intact files of the same call byte-exact -- what the damaged ones look like is undefined (Pillow's output for
    them depends on its own error recovery and is not compared)"""
    rng = np.random.RandomState(5)
    good = [_encode(_picture(200, 150, 1), quality=85, subsampling=2), _encode(_picture(97, 61, 2), quality=70, subsampling=0)]
    bad = []
    for blob in (good[0], good[1], _encode(_picture(160, 120, 3), quality=90, subsampling=1, restart_marker_rows=2)):
        b = bytearray(blob)
        start = blob.index(b"\xff\xda") + 14
        for _ in range(40):
            k = rng.randint(start, len(b) - 2)
            if b[k] != 0xFF and b[k - 1] != 0xFF:
                b[k] = rng.randint(0, 255)
        bad.append(bytes(b))
    out = jpeg.decode_many([good[0]] + bad + [good[1]], "cuda:0")
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), _reference(good[0])) and np.array_equal(out[-1].cpu().numpy(), _reference(good[1]))
    for t, blob in zip(out[1:-1], bad):
        p = jpeg.parse(blob)
        assert tuple(t.shape) == (p.info.height, p.info.width, 3)


def test_images_from_list_mirror(tmp_path):
    """gandtr_amd.datasets.ImagesFromList against the reference's per-item steps (genericdataset.py:58-99) done with Pillow / numpy: load, crop
    to the bounding box, imresize scaled by the crop's share of the full image, totensor | normalize"""
    from gandtr_amd.datasets import ImagesFromList
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    names, bbxs = [], []
    for i, (w, h) in enumerate([(400, 300), (280, 360), (333, 222)]):
        names.append("im%d.jpg" % i)
        _picture(w, h, 30 + i).save(str(tmp_path / names[-1]), "JPEG", quality=88, subsampling=i % 3)
        bbxs.append(None if i == 1 else (20.4 + i, 31.6, w - 40.2, h - 25.5))
    ds = ImagesFromList(str(tmp_path), names, imsize=200, bbxs=bbxs, transform=(mean, std), device="cuda:0")
    assert len(ds) == 3
    got = ds.batch(range(3))
    for i, g in enumerate(got):
        img = Image.open(str(tmp_path / names[i])).convert("RGB")
        full = max(img.size)
        if bbxs[i]:
            img = img.crop(bbxs[i])
            img.thumbnail((200 * max(img.size) / full,) * 2, Image.LANCZOS)
        else:
            img.thumbnail((200, 200), Image.LANCZOS)
        want = (np.asarray(img).astype(np.float32) / 255.0 - np.array(mean, np.float32)) / np.array(std, np.float32)
        assert tuple(g.shape) == (3, want.shape[0], want.shape[1]), (i, g.shape, want.shape)
        assert np.abs(g.cpu().numpy() - want.transpose(2, 0, 1)).max() < 1e-5
    assert torch.equal(ds[1], got[1])
    plain = ImagesFromList("", [str(tmp_path / names[0])], device="cuda:0")          # no imsize, no transform: the decoded pixels
    assert np.array_equal(plain[0].cpu().numpy(), np.asarray(Image.open(str(tmp_path / names[0])).convert("RGB")))
    with pytest.raises(RuntimeError, match="0 images"):
        ImagesFromList("", [])
