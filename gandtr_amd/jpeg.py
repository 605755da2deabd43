"""JPEG decoding on the device: the first half of the ingest row.

Reference (per image, on the CPU): ``pil_loader`` = ``Image.open(f).convert('RGB')`` (mdir/external/cirtorch/datasets/datahelpers.py:39-47),
called from ``ImagesFromList.__getitem__`` (genericdataset.py:66-102) before ``imresize`` and the transform.  Here the FILE bytes are
uploaded and decoded by HIP launches (gandtr_amd/csrc/jpeg.hip): Huffman decoding in parallel within each file, DC prediction,
dequantisation + inverse DCT, chroma upsampling and colour conversion -- bit-identical to Pillow / libjpeg-turbo.  The host parses the
headers and strips the byte stuffing from the entropy-coded segment while staging it for the upload; it decodes nothing.

Baseline files (8-bit, Huffman, one interleaved scan, grayscale or YCbCr 4:4:4 / 4:2:2 / 4:2:0) are decoded on the device end to end.
Progressive files (SOF2: their scans refine the same coefficients several times over, a sequential pass per scan) have their coefficients
decoded by the library on a few host threads and go through the device for everything behind the entropy decoder -- same arithmetic, same
bytes.  ``parse`` raises ValueError for anything else (arithmetic coding, CMYK, 12-bit, ...), with the reason; ``load_many`` passes that on
unless the caller supplies its own loader for such files."""
import ctypes
import os

import numpy as np
import torch

from . import _hip


class Parsed:
    """Headers of one baseline JPEG file (``info`` = struct gdt_jpeg_info) plus the file bytes."""

    def __init__(self, data):
        if not isinstance(data, (bytes, bytearray, memoryview)):
            raise TypeError("JPEG data must be bytes-like")
        self.data = bytes(data)
        self.info = _hip.JpegInfo()
        _hip.check(_hip.load().gdt_jpeg_parse(self.data, len(self.data), ctypes.byref(self.info)))

    @property
    def size(self):
        return self.info.width, self.info.height

    @property
    def mode(self):
        return "L" if self.info.ncomp == 1 else "RGB"


def parse(data):
    return Parsed(data)


_THREADS = max(1, min(8, (os.cpu_count() or 2) // 2))


def _ptr_array(blobs):
    keep = [ctypes.c_char_p(b) for b in blobs]                # (borrowed pointers into the bytes objects, kept alive by the caller's list)
    return (ctypes.c_void_p * len(blobs))(*[ctypes.cast(k, ctypes.c_void_p).value for k in keep]), keep


def parse_many(blobs):
    """``[Parsed(b) for b in blobs]`` in ONE library call on a few host threads (gdt_jpeg_parse_batch); the first refused file raises the
    ValueError ``parse`` would raise for it."""
    blobs = [bytes(b) for b in blobs]
    n = len(blobs)
    if n == 0:
        return []
    infos = (_hip.JpegInfo * n)()
    status = (ctypes.c_int * n)()
    files, keep = _ptr_array(blobs)
    sizes = (ctypes.c_size_t * n)(*[len(b) for b in blobs])
    _hip.check(_hip.load().gdt_jpeg_parse_batch(files, sizes, n, infos, status, _THREADS))
    out = []
    for i, b in enumerate(blobs):
        if status[i] != _hip.GDT_OK:
            Parsed(b)                                          # raises with the reason
            raise ValueError("JPEG file %d of the list was refused" % i)
        p = Parsed.__new__(Parsed)
        p.data, p.info = b, infos[i]
        out.append(p)
    return out


import threading

_stages = {}                    # (thread id, device index) -> [pinned buffer, event recorded behind its last upload]
_stages_lock = threading.Lock()


def _staging(nbytes, device):
    """Pinned upload buffer of THIS thread for ``device``, grown geometrically and kept (allocating pinned memory costs more than decoding a
    small file).  Before it is rewritten the event recorded behind the previous upload from it is awaited -- whichever stream that upload
    ran on -- so a caller on another stream, device or thread can never overwrite bytes that are still in flight."""
    key = (threading.get_ident(), device.index if device.index is not None else torch.cuda.current_device())
    with _stages_lock:
        entry = _stages.get(key)
        if entry is None or entry[0].numel() < nbytes:
            if entry is not None and entry[1] is not None:
                entry[1].synchronize()
            entry = [torch.empty(max(nbytes, 1 << 20) * 3 // 2, dtype=torch.uint8).pin_memory(), None]
            _stages[key] = entry
    if entry[1] is not None:
        entry[1].synchronize()
    return entry[0][:nbytes], entry


def _decode_progressive(parsed, device):
    """progressive files: coefficients by the library's host decoder (one call, a few threads), one upload, device back end"""
    lib = _hip.load()
    n = len(parsed)
    offs, total = [], 0
    for p in parsed:
        offs.append(total)
        total += p.info.mcus_x * p.info.mcus_y * p.info.blocks_per_mcu * 64
    coef = torch.empty(total, dtype=torch.int16).pin_memory()
    infos = (_hip.JpegInfo * n)(*[p.info for p in parsed])
    files, keep = _ptr_array([p.data for p in parsed])
    status = (ctypes.c_int * n)()
    coef_off = (ctypes.c_size_t * n)(*offs)
    _hip.check(lib.gdt_jpeg_progressive_coefficients_batch(files, (ctypes.c_size_t * n)(*[len(p.data) for p in parsed]), infos, n, coef.data_ptr(),
                                                           coef_off, status, _THREADS))
    for i in range(n):
        if status[i] != _hip.GDT_OK:            # raise with the reason
            _hip.check(lib.gdt_jpeg_progressive_coefficients(parsed[i].data, len(parsed[i].data), ctypes.byref(infos[i]), coef.data_ptr() + 2 * offs[i]))
            raise ValueError("JPEG file %d of the list was refused" % i)
    with torch.cuda.device(device):
        coef_dev = coef.to(device, non_blocking=True)
        out_offs, out_total = [], 0
        for p in parsed:
            out_offs.append(out_total)
            out_total += (p.info.width * p.info.height * 3 + 255) // 256 * 256
        out = torch.empty(out_total, dtype=torch.uint8, device=device)
        dst = (ctypes.c_void_p * n)(*[out.data_ptr() + oo for oo in out_offs])
        nbytes = ctypes.c_size_t()
        _hip.check(lib.gdt_jpeg_decode_coef_workspace_bytes(infos, n, ctypes.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        _hip.check(lib.gdt_jpeg_decode_coef_u8_batch(infos, coef_dev.data_ptr(), coef_off, dst, n, ws.data_ptr(), nbytes.value,
                                                     torch.cuda.current_stream().cuda_stream))
        torch.cuda.current_stream().synchronize()      # (the pinned coefficient buffer is released with this frame)
    return [out[oo:oo + p.info.width * p.info.height * 3].view(p.info.height, p.info.width, 3) for p, oo in zip(parsed, out_offs)]


def decode_many(blobs, device=None, sequential=False):
    """Decodes a list of JPEG files (bytes, or ``Parsed``) in one library call per kind (baseline / progressive).  Returns uint8 H x W x 3
    tensors on the device, equal to ``np.asarray(Image.open(f).convert('RGB'))``.  ``sequential`` selects the one-thread-per-restart-interval
    entropy decoder of the baseline path (the checker of the parallel one).  The call synchronises the stream (pinned staging buffers; the
    repair rounds of the parallel entropy decoder read a flag) and is not capturable into a hipGraph."""
    lib = _hip.load()
    if not blobs:
        return []
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.type != "cuda":
        raise ValueError("JPEG decoding runs on a HIP device only (there is no CPU path)")
    raw = [i for i, b in enumerate(blobs) if not isinstance(b, Parsed)]
    parsed = list(blobs)
    for i, p in zip(raw, parse_many([blobs[i] for i in raw])):
        parsed[i] = p
    prog = [i for i, p in enumerate(parsed) if p.info.progressive]
    if prog:
        out = [None] * len(parsed)
        for i, t in zip(prog, _decode_progressive([parsed[i] for i in prog], device)):
            out[i] = t
        base = [i for i in range(len(parsed)) if not parsed[i].info.progressive]
        if base:
            for i, t in zip(base, decode_many([parsed[i] for i in base], device, sequential)):
                out[i] = t
        return out
    n = len(parsed)
    # staging: every image's un-stuffed scan at a 16-byte aligned offset of ONE pinned buffer -> one upload
    offs, total, seg_index, nseg = [], 0, [], 0
    for p in parsed:
        offs.append(total)
        total += (int(p.info.scan_capacity) + 15) // 16 * 16
        seg_index.append(nseg)
        nseg += p.info.nsegments + 1
    stage, stage_entry = _staging(total, device)
    seg_all = (ctypes.c_uint * nseg)()
    infos = (_hip.JpegInfo * n)(*[p.info for p in parsed])
    files, keep = _ptr_array([p.data for p in parsed])
    _hip.check(lib.gdt_jpeg_extract_scan_batch(files, (ctypes.c_size_t * n)(*[len(p.data) for p in parsed]), infos, n, stage.data_ptr(),
                                               (ctypes.c_size_t * n)(*offs), seg_all, (ctypes.c_size_t * n)(*seg_index), _THREADS))
    seg_base = ctypes.addressof(seg_all)
    with torch.cuda.device(device):
        scans = stage.to(device, non_blocking=True)
        stage_entry[1] = torch.cuda.Event()
        stage_entry[1].record(torch.cuda.current_stream())
        out_offs, out_total = [], 0
        for p in parsed:
            out_offs.append(out_total)
            out_total += (p.info.width * p.info.height * 3 + 255) // 256 * 256
        out = torch.empty(out_total, dtype=torch.uint8, device=device)
        items = (_hip.JpegItem * len(parsed))()
        for i, (p, off, oo) in enumerate(zip(parsed, offs, out_offs)):
            items[i].info = ctypes.pointer(infos[i])
            items[i].scan = scans.data_ptr() + off
            items[i].seg_off = ctypes.cast(seg_base + 4 * seg_index[i], ctypes.POINTER(ctypes.c_uint))
            items[i].dst_hwc = out.data_ptr() + oo
        nbytes = ctypes.c_size_t()
        _hip.check(lib.gdt_jpeg_decode_workspace_bytes(items, len(parsed), ctypes.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        _hip.check(lib.gdt_jpeg_decode_u8_batch(items, len(parsed), 1 if sequential else 0, ws.data_ptr(), nbytes.value,
                                                torch.cuda.current_stream().cuda_stream))
    return [out[oo:oo + p.info.width * p.info.height * 3].view(p.info.height, p.info.width, 3) for p, oo in zip(parsed, out_offs)]


def decode(blob, device=None):
    return decode_many([blob], device)[0]


def _read(source):
    if isinstance(source, (bytes, bytearray, memoryview)):
        return bytes(source)
    with open(source, "rb") as handle:
        return handle.read()


def load_many(sources, device=None, host_loader=None):
    """Device-side ``pil_loader`` over a list of paths / file contents: decoded uint8 H x W x 3 tensors on the device, in input order.
    Baseline and progressive JPEG files go through ``decode_many`` (one call per kind).  A file the decoder does not take (PNG,
    arithmetic-coded or CMYK JPEG, ...) raises the ValueError of ``parse`` -- unless the caller passes ``host_loader``, a callable
    ``bytes -> H x W x 3 uint8 array`` (the reference's own ``pil_loader`` is the natural one); its result is uploaded as is.
    Nothing in this module decodes on the host by itself."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    blobs = [_read(s) for s in sources]
    parsed, on_device, rest = [], [], []
    for i, b in enumerate(blobs):
        try:
            parsed.append(Parsed(b))
            on_device.append(i)
        except ValueError:
            if host_loader is None:
                raise
            rest.append(i)
    out = [None] * len(blobs)
    for i, t in zip(on_device, decode_many(parsed, device)):
        out[i] = t
    for i in rest:
        out[i] = torch.from_numpy(np.ascontiguousarray(host_loader(blobs[i]), dtype=np.uint8)).to(device)
    return out


def ingest_files(sources, imsize, mean, std, clahe_clip=None, clahe_grid=8, device=None, host_loader=None):
    """``ImagesFromList.__getitem__`` (genericdataset.py:66-102) for a list of files, on the device end to end: decode, ``imresize`` to
    ``imsize`` (Pillow's thumbnail arithmetic), [0, 1] scaling, optional CLAHE, normalisation.  Returns fp32 3 x h x w tensors."""
    from . import ingest
    return ingest.ingest_many(load_many(sources, device, host_loader), imsize, mean, std, clahe_clip, clahe_grid)
