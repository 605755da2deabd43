"""Host half of the device JPEG decoder (header parsing, un-stuffing, restart intervals): pure C, no device needed."""
import ctypes
import io

import numpy as np
import pytest

from gandtr_amd import _hip, jpeg

Image = pytest.importorskip("PIL.Image")


def _encode(arr, **kw):
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, "JPEG", **kw)
    return buf.getvalue()


def _scan(p):
    out = (ctypes.c_ubyte * int(p.info.scan_capacity))()
    seg = (ctypes.c_uint * (p.info.nsegments + 1))()
    _hip.check(_hip.load().gdt_jpeg_extract_scan(p.data, len(p.data), ctypes.byref(p.info), out, seg))
    return bytes(out), list(seg)


def test_headers_agree_with_pillow():
    rng = np.random.RandomState(0)
    for (h, w), sub, q in [((33, 47), 2, 90), ((8, 8), 0, 50), ((100, 3), 1, 75), ((64, 64), 2, 20)]:
        blob = _encode(rng.randint(0, 256, (h, w, 3), dtype=np.uint8), quality=q, subsampling=sub)
        p = jpeg.parse(blob)
        with Image.open(io.BytesIO(blob)) as img:
            assert p.size == img.size and p.mode == img.mode
            assert (p.info.hs[0], p.info.vs[0]) == {0: (1, 1), 1: (2, 1), 2: (2, 2)}[sub]
            # quantisation tables: Pillow reports them in zigzag order of the file; info holds the natural order
            zig = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
                   56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
            for tq, table in img.quantization.items():
                # (Pillow >= 8.3 de-zigzags: accept either convention, exactly)
                nat = list(p.info.quant[tq])
                assert list(table) == nat or [nat[zig[i]] for i in range(64)] == list(table)
        assert p.info.mcus_x == -(-w // (8 * p.info.hs[0])) and p.info.mcus_y == -(-h // (8 * p.info.vs[0]))
        assert p.info.nsegments == 1 and p.info.restart_interval == 0
    gray = jpeg.parse(_encode(rng.randint(0, 256, (20, 30), dtype=np.uint8), quality=70))
    assert gray.mode == "L" and gray.info.blocks_per_mcu == 1 and gray.info.mcus_x == 4 and gray.info.mcus_y == 3


def test_scan_extraction_removes_stuffing_and_cuts_at_restart_markers():
    rng = np.random.RandomState(1)
    arr = rng.randint(0, 256, (96, 160, 3), dtype=np.uint8)             # noise at quality 100: plenty of 0xFF bytes in the scan
    plain = jpeg.parse(_encode(arr, quality=100, subsampling=0))
    raw = plain.data[int(plain.info.scan_offset):]
    stuffed = raw.count(b"\xff\x00")
    assert stuffed > 10
    data, seg = _scan(plain)
    assert seg[0] == 0 and seg[1] == len(raw) - 2 - stuffed                 # (minus the EOI marker and the stuffing zeros)
    assert data[:seg[1]] == raw[:-2].replace(b"\xff\x00", b"\xff") and set(data[seg[1]:]) == {0}
    rst = jpeg.parse(_encode(arr, quality=90, subsampling=2, restart_marker_rows=1))
    assert rst.info.restart_interval == rst.info.mcus_x and rst.info.nsegments == rst.info.mcus_y
    data, seg = _scan(rst)
    assert len(seg) == rst.info.nsegments + 1 and all(a < b for a, b in zip(seg, seg[1:]))
    # the same picture without restart markers carries the same coefficients: equal size up to the padding bits of each interval
    one = jpeg.parse(_encode(arr, quality=90, subsampling=2))
    assert abs(_scan(one)[1][1] - seg[-1]) <= rst.info.nsegments


def test_unsupported_kinds_are_refused_with_the_reason():
    rng = np.random.RandomState(2)
    arr = rng.randint(0, 256, (40, 40, 3), dtype=np.uint8)
    prog = jpeg.parse(_encode(arr, quality=80, progressive=True))          # SOF2: accepted; its scans go through the coefficient decoder
    assert prog.info.progressive == 1 and prog.size == (40, 40)
    buf = io.BytesIO()
    Image.fromarray(arr).convert("CMYK").save(buf, "JPEG")
    cmyk = jpeg.parse(buf.getvalue())                                     # four components (Adobe transform 0): decoded and converted like Pillow's convert('RGB')
    assert cmyk.info.ncomp == 4 and cmyk.info.adobe_transform == 0 and cmyk.info.blocks_per_mcu == 4 and cmyk.mode == "RGB"
    buf = io.BytesIO()
    Image.fromarray(arr).convert("CMYK").save(buf, "JPEG", subsampling=2)
    assert jpeg.parse(buf.getvalue()).info.blocks_per_mcu == 7            # first component 2 x 2, the others 1 x 1
    two = bytearray(_encode(arr, quality=80))                             # a frame that claims two components
    sof = two.index(b"\xff\xc0")
    two[sof + 9] = 2
    with pytest.raises(ValueError):
        jpeg.parse(bytes(two))
    with pytest.raises(ValueError, match="SOI"):
        jpeg.parse(b"\x89PNG\r\n\x1a\n" + bytes(64))
    good = _encode(arr, quality=80)
    with pytest.raises(ValueError):
        jpeg.parse(good[:200])                                            # truncated inside the headers
    with pytest.raises(TypeError):
        jpeg.parse("not bytes")
    with pytest.raises(ValueError, match="HIP device"):
        jpeg.decode_many([good], "cpu")


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "jpeg_cases.npz"))


def test_golden_files_parse_and_pillow_still_reproduces_them():
    """tests/golden/jpeg_cases.npz (make_jpeg_golden.py): files + the reference loader's output.  Here: every file is accepted by the
    host parser with the geometry of the stored output, and the installed Pillow decodes it to exactly the stored pixels (so a
    different libjpeg build would show up here, not as a mystery on the GPU)."""
    g = _golden()
    assert bool(g["libjpeg_turbo"])
    for name in g["names"]:
        blob, want = bytes(g["file_" + str(name)]), g["rgb_" + str(name)]
        p = jpeg.parse(blob)
        assert p.size == (want.shape[1], want.shape[0])
        with Image.open(io.BytesIO(blob)) as img:
            assert np.array_equal(np.asarray(img.convert("RGB")), want), name


def test_every_truncation_pillow_refuses_is_refused():
    import ctypes
    from gandtr_amd import _hip
    """the reference's loader (pil_loader = Image.open(f).convert('RGB'), datahelpers.py:39-47) raises OSError for a file cut inside its
    scan; the device path must not hand back a zero-padded image for it (datasets.ImagesFromList(ignore_errors=...) and
    extract_vectors_from_files would embed it): whatever Pillow refuses, gdt_jpeg_parse refuses"""
    g = _golden()
    rng = np.random.RandomState(5)
    blobs = [bytes(g["file_" + str(n)]) for n in g["names"]]
    blobs.append(_encode(rng.randint(0, 256, (64, 64, 3), dtype=np.uint8), quality=85))
    pillow_refused = ours_refused = 0
    for blob in blobs:
        sos = blob.index(b"\xff\xda")
        cuts = sorted(set([sos + 14, (sos + len(blob)) // 2, 2 * len(blob) // 3, len(blob) - 40, len(blob) - 3, len(blob) - 2, len(blob) - 1]))
        for k in cuts:
            if not sos < k < len(blob):
                continue
            cut = blob[:k]
            try:
                with Image.open(io.BytesIO(cut)) as img:
                    img.convert("RGB")
                pillow_ok = True
            except OSError:
                pillow_ok = False
            try:
                p = jpeg.parse(cut)
                ours_ok = True
                if p.info.progressive:                     # (the scans of a progressive file are walked by its coefficient decoder)
                    coef = np.zeros(p.info.mcus_x * p.info.mcus_y * p.info.blocks_per_mcu * 64, dtype=np.int16)
                    ours_ok = _hip.load().gdt_jpeg_progressive_coefficients(cut, len(cut), ctypes.byref(p.info), coef.ctypes.data) == 0
            except ValueError:
                ours_ok = False
            pillow_refused += not pillow_ok
            ours_refused += not ours_ok
            assert not (ours_ok and not pillow_ok), (len(blob), k)
    assert pillow_refused > 20 and ours_refused >= pillow_refused


def test_parser_survives_truncated_and_corrupted_files():
    """the host parser reads untrusted bytes: every truncation and a few thousand random corruptions of valid files must come back as a
    clean ValueError or as a consistent info (whose scan extraction then stays inside the buffer it was sized for) -- never a crash"""
    rng = np.random.RandomState(3)
    arr = rng.randint(0, 256, (40, 56, 3), dtype=np.uint8)
    files = [_encode(arr, quality=85, subsampling=2), _encode(arr, quality=60, subsampling=0, optimize=True, restart_marker_blocks=2),
             _encode(arr[:, :, 0], quality=75)]
    accepted = refused = 0
    for blob in files:
        header = blob.index(b"\xff\xda") + 16
        cases = [blob[:k] for k in range(0, len(blob), 1 if len(blob) < 1200 else 3)]
        for _ in range(1500):
            b = bytearray(blob)
            for _ in range(rng.randint(1, 4)):
                b[rng.randint(2, header)] = rng.randint(0, 256)
            cases.append(bytes(b))
        for c in cases:
            try:
                p = jpeg.parse(c)
            except ValueError:
                refused += 1
                continue
            accepted += 1
            assert 0 < p.info.width <= 65535 and 0 < p.info.height <= 65535 and p.info.nsegments >= 1
            data, seg = _scan(p)
            assert len(data) == p.info.scan_capacity and seg[-1] + 16 <= p.info.scan_capacity and all(a <= b2 for a, b2 in zip(seg, seg[1:]))
    assert accepted > 50 and refused > 1000


def test_images_from_list_constructor_contract():
    """the dataset mirror refuses what it does not mirror and keeps the reference's empty-list error (genericdataset.py:54-55); no device needed"""
    from gandtr_amd.datasets import ImagesFromList
    with pytest.raises(RuntimeError, match="0 images"):
        ImagesFromList("", [], device="cpu")
    with pytest.raises(NotImplementedError):
        ImagesFromList("store.h5", ["a.jpg"], device="cpu")
    with pytest.raises(NotImplementedError):
        ImagesFromList("", ["a.jpg"], load_images_with_bbx=True, device="cpu")
    ds = ImagesFromList("/data", ["a.jpg", "sub/b.jpg"], imsize=362, device="cpu")
    assert len(ds) == 2 and ds.images_fn == ["/data/a.jpg", "/data/sub/b.jpg"] and "Number of images: 2" in repr(ds)
    with pytest.raises(OSError):
        ds.batch([0])                                   # a missing file is the loader's error, as in the reference (genericdataset.py:70-75)
    blob = _encode(np.zeros((16, 16, 3), np.uint8), quality=80)
    with pytest.raises(ValueError, match="HIP device"):
        ImagesFromList("", [blob], device="cpu").batch([0])      # decoding needs the device: there is no host path


def test_progressive_coefficient_decoder_on_the_host():
    """gdt_jpeg_progressive_coefficients (pure host code): the quantised coefficients of a progressive file equal those of the BASELINE file Pillow
    writes from the same pixels at the same quality (the two codings carry the same DCT output; libjpeg computes it identically), checked through
    Pillow's own dequantised view: both files decode to the same pixels -- and here structurally: same DC terms, same non-zero count."""
    import ctypes
    from gandtr_amd import _hip
    lib = _hip.load()
    rng = np.random.RandomState(7)
    base = rng.randint(0, 256, (12, 16, 3), dtype=np.uint8)
    arr = np.asarray(Image.fromarray(base).resize((96, 72), Image.BICUBIC))
    for sub in (0, 2):
        blob = _encode(arr, quality=85, subsampling=sub, progressive=True)
        p = jpeg.parse(blob)
        n = p.info.mcus_x * p.info.mcus_y * p.info.blocks_per_mcu
        coef = np.zeros(n * 64, dtype=np.int16)
        assert lib.gdt_jpeg_progressive_coefficients(blob, len(blob), ctypes.byref(p.info), coef.ctypes.data) == 0
        coef = coef.reshape(n, 64)
        # DC of the first luma block = round(mean of the level-shifted block * 8 / q00): within one quantisation step of the pixels' block mean
        with Image.open(io.BytesIO(blob)) as img:
            y = np.asarray(img.convert("YCbCr"))[:8, :8, 0].astype(np.float64)
        q00 = p.info.quant[p.info.tq[0]][0]
        assert abs(coef[0, 0] * q00 / 8.0 - (y.mean() - 128.0)) <= q00 / 8.0 + 1.5
        assert (coef != 0).sum() > n                                        # AC bands were decoded, too
        # every truncation inside the scans is refused (Pillow raises for them as well)
        for cut in (len(blob) // 2, len(blob) - 2):
            rc = lib.gdt_jpeg_progressive_coefficients(blob[:cut], cut, ctypes.byref(p.info), coef.ctypes.data)
            assert rc != 0 and b"truncated" in lib.gdt_last_error()


def _dht_segments(blob):
    """(offset of the 0xFF of every DHT marker segment, its length field) in file order"""
    out, pos = [], 2
    while pos + 4 <= len(blob):
        assert blob[pos] == 0xFF
        m = blob[pos + 1]
        if m == 0xD9:
            break
        ln = (blob[pos + 2] << 8) | blob[pos + 3]
        if m == 0xC4:
            out.append((pos, ln))
        if m == 0xDA:                                   # skip the entropy-coded data up to the next marker that is not RSTn / stuffing
            pos += 2 + ln
            while pos + 1 < len(blob) and not (blob[pos] == 0xFF and blob[pos + 1] != 0 and not 0xD0 <= blob[pos + 1] <= 0xD7 and blob[pos + 1] != 0xFF):
                pos += 1
            continue
        pos += 2 + ln
    return out


def test_oversubscribed_huffman_tables_are_refused():
    """a DHT whose 16 length counts do not form a prefix code (e.g. all symbols at length 1) used to pass gdt_jpeg_parse and
    then index far past the 512-entry look-up table of the host progressive decoder (stack smash).  The counts are now checked the way the
    library does (codes of length l fit l bits, none all ones): tables 0 / 1 and 2 / 3, before the first scan (parser) and between the scans
    (coefficient decoder), baseline files too."""
    lib = _hip.load()
    rng = np.random.RandomState(11)
    arr = rng.randint(0, 256, (48, 64, 3), dtype=np.uint8)

    def smash(blob, seg_index, th=None):
        b = bytearray(blob)
        pos, ln = _dht_segments(blob)[seg_index]
        q = pos + 4                                       # first table of the segment: Tc/Th, 16 counts, symbols
        total = sum(b[q + 1:q + 17])
        assert 2 <= total <= 255
        b[q + 1:q + 17] = bytes([total] + [0] * 15)      # every symbol a code of length 1
        if th is not None:
            b[q] = (b[q] & 0xF0) | th
        return bytes(b)

    base = _encode(arr, quality=80, subsampling=2)
    prog = _encode(arr, quality=80, subsampling=2, progressive=True)
    assert len(_dht_segments(base)) >= 1 and len(_dht_segments(prog)) >= 2
    first_scan = prog.index(b"\xff\xda")
    assert any(pos < first_scan for pos, _ in _dht_segments(prog)) and any(pos > first_scan for pos, _ in _dht_segments(prog)), \
        "Pillow's progressive files define tables before the first scan and between the scans"

    def ours(blob):
        """None if the file decodes on the host side, else the error text (a crash would end the test run)"""
        try:
            p = jpeg.parse(blob)
        except ValueError as e:
            return str(e)
        if not p.info.progressive:
            return None
        coef = np.zeros(p.info.mcus_x * p.info.mcus_y * p.info.blocks_per_mcu * 64, dtype=np.int16)
        rc = lib.gdt_jpeg_progressive_coefficients(blob, len(blob), ctypes.byref(p.info), coef.ctypes.data)
        return None if rc == 0 else lib.gdt_last_error().decode()

    def pillow_ok(blob):
        try:
            Image.open(io.BytesIO(blob)).convert("RGB")
            return True
        except Exception:
            return False

    # a table a scan USES: refused with the reason, as the reference's loader refuses it (the library checks a table when a scan starts)
    used = [smash(base, i) for i in range(len(_dht_segments(base)))] + [smash(prog, i) for i in range(len(_dht_segments(prog)))]
    for blob in used:
        err = ours(blob)
        assert err is not None and "Huffman" in err, err
        assert not pillow_ok(blob)
    # the same broken tables under the numbers 2 / 3 (progressive files may use them; here no scan does, or the scan loses its own table): no crash,
    # and nothing is accepted that the reference's loader refuses
    for i in range(len(_dht_segments(prog))):
        for th in (2, 3):
            blob = smash(prog, i, th)
            assert not (ours(blob) is None and not pillow_ok(blob))
    # ... and USED under those numbers: scan headers re-pointed to table 2 with the DHT that defines it
    pos, ln = _dht_segments(prog)[0]
    b = bytearray(smash(prog, 0, th=2))
    tc = b[pos + 4] >> 4
    sos = prog.index(b"\xff\xda")
    ns = b[sos + 4]
    for c in range(ns):                                   # first scan (DC, interleaved): its table selectors
        sel = b[sos + 6 + 2 * c]
        b[sos + 6 + 2 * c] = (sel & 0x0F) | 0x20 if tc == 0 else (sel & 0xF0) | 0x02
    err = ours(bytes(b))
    assert err is not None and "Huffman" in err, err
    # a complete code with an all-ones codeword (2 symbols of length 1) is refused as the library refuses it
    b = bytearray(base)
    pos, _ = _dht_segments(base)[0]
    b[pos + 5:pos + 21] = bytes([2] + [0] * 15)
    err = ours(bytes(b))
    assert err is not None and not pillow_ok(bytes(b))


def test_progressive_four_component_restart_resets_all_predictors():
    """the DC predictor of the FOURTH component was not reset at restart markers of a progressive scan.  The same CMYK picture
    coded with and without restart intervals carries the same coefficients."""
    lib = _hip.load()
    rng = np.random.RandomState(5)
    small = rng.randint(0, 256, (9, 11, 4), dtype=np.uint8)
    img = Image.fromarray(small, "CMYK").resize((44, 36), Image.BICUBIC)

    def coefs(**kw):
        buf = io.BytesIO()
        img.save(buf, "JPEG", quality=84, progressive=True, **kw)
        blob = buf.getvalue()
        p = jpeg.parse(blob)
        assert p.info.ncomp == 4 and p.info.progressive
        c = np.zeros(p.info.mcus_x * p.info.mcus_y * p.info.blocks_per_mcu * 64, dtype=np.int16)
        assert lib.gdt_jpeg_progressive_coefficients(blob, len(blob), ctypes.byref(p.info), c.ctypes.data) == 0, lib.gdt_last_error()
        return c.reshape(-1, p.info.blocks_per_mcu, 64), p.info.restart_interval

    plain, r0 = coefs()
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1), dict(restart_marker_blocks=1)):
        with_rst, r1 = coefs(**kw)
        assert r0 == 0 and r1 > 0
        assert np.array_equal(plain, with_rst), [int((plain[:, c] != with_rst[:, c]).sum()) for c in range(plain.shape[1])]
