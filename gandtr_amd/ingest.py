"""Image ingest on the device after decoding.

Reference (per image, on the CPU through Pillow / numpy): ``imresize`` = ``img.thumbnail((s, s), LANCZOS)``
(mdir/external/cirtorch/datasets/datahelpers.py:75-82, called from genericdataset.py:66-102) followed by the hub transform
``pil2np | apply_clahe:1.0 | totensor | normalize`` (mdir/hub/embedding.yml:14, core_transforms.py:35-100).
Here the decoded uint8 H x W x 3 image is uploaded once; resize (bit-identical to Pillow), [0, 1] scaling, optional CLAHE and
normalisation run as HIP launches (gandtr_amd/csrc/ingest.hip, clahe.hip).  The size / reducing-gap plan below mirrors Pillow's
Python layer (PIL/Image.py: thumbnail, resize); JPEG decoding stays on the host.  No CPU fallback."""
import ctypes
import math

import torch

from . import _hip


def thumbnail_size(width, height, imsize):
    """size Image.thumbnail((imsize, imsize)) resizes to, or None when the image already fits (PIL/Image.py preserve_aspect_ratio)"""
    x = y = math.floor(imsize)
    if x >= width and y >= height:
        return None

    def round_aspect(number, key):
        return max(min(math.floor(number), math.ceil(number), key=key), 1)

    aspect = width / height
    if x / y >= aspect:
        x = round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
    else:
        y = round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
    return x, y


def reduce_plan(width, height, out_w, out_h, reducing_gap=2.0):
    """Image.resize(..., reducing_gap): integer box-reduction factors and the source box of the resampling step (in pixels of the
    reduced image).  thumbnail passes the full image as the box, for which Pillow's safe box is the full image as well."""
    fx = int(width / out_w / reducing_gap) or 1
    fy = int(height / out_h / reducing_gap) or 1
    if fx > 1 or fy > 1:
        return fx, fy, (0.0, 0.0, width / fx, height / fy)
    return 1, 1, (0.0, 0.0, float(width), float(height))


def _f(vals, n):
    if vals is None:
        return None
    vals = [float(v) for v in vals]
    if len(vals) != n:
        raise ValueError("expected %d per-channel values, got %d" % (n, len(vals)))
    return (ctypes.c_float * n)(*vals)


def resize(img, out_w, out_h, factors=(1, 1), box=None, want_u8=True, mean_std=None, want_chw=False):
    """Pillow's reduce(factors) + resize((out_w, out_h), LANCZOS, box) on a uint8 H x W x C device tensor.
    Returns (uint8 out_h x out_w x C or None, fp32 C x out_h x out_w or None)."""
    lib = _hip.load()
    if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3:
        raise ValueError("ingest needs a uint8 H x W x C tensor on a HIP device")
    img = img.contiguous()
    h, w, c = img.shape
    fx, fy = factors
    dst = torch.empty((out_h, out_w, c), dtype=torch.uint8, device=img.device) if want_u8 else None
    chw = torch.empty((c, out_h, out_w), dtype=torch.float32, device=img.device) if want_chw else None
    mean, std = mean_std if mean_std is not None else (None, None)
    need = ctypes.c_size_t()
    with torch.cuda.device(img.device):
        _hip.check(lib.gdt_ingest_workspace_bytes(h, w, c, fx, fy, out_w, out_h, ctypes.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=img.device)
        _hip.check(lib.gdt_ingest_resize_u8(img.data_ptr(), h, w, c, fx, fy, _f(box, 4), out_w, out_h,
                                            dst.data_ptr() if want_u8 else None, chw.data_ptr() if want_chw else None,
                                            _f(mean, c), _f(std, c), ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream(img.device).cuda_stream))
    return dst, chw


def _plan(img, imsize):
    h, w, _ = img.shape
    size = thumbnail_size(w, h, imsize) if imsize is not None else None
    if size is None:
        size = (w, h)
    fx, fy, box = reduce_plan(w, h, size[0], size[1]) if size != (w, h) else (1, 1, None)
    return size, (fx, fy), box


def imresize(img, imsize):
    """datahelpers.imresize: ``img.thumbnail((imsize, imsize), LANCZOS)`` on a decoded uint8 H x W x C device tensor."""
    size, factors, box = _plan(img, imsize)
    if size == (img.shape[1], img.shape[0]):
        return img
    return resize(img, size[0], size[1], factors, box)[0]


def ingest(img, imsize, mean, std, clahe_clip=None, clahe_grid=8):
    """decoded uint8 H x W x 3 (device) -> fp32 3 x h x w, what ``imresize`` followed by
    ``pil2np | [apply_clahe:clip] | totensor | normalize`` produce (hub transform: mdir/hub/embedding.yml:14)."""
    size, factors, box = _plan(img, imsize)
    if clahe_clip is None:
        return resize(img, size[0], size[1], factors, box, want_u8=False, mean_std=(mean, std), want_chw=True)[1]
    from . import clahe
    unit = resize(img, size[0], size[1], factors, box, want_u8=False, want_chw=True)[1]          # [0, 1] RGB planes
    return clahe.clahe_lab(unit[None], clahe_clip, clahe_grid, None, (mean, std))[0]


def resize_many(images, plans, want_u8=True, mean_std=None, want_chw=False):
    """``resize`` for a list of uint8 H x W x C device tensors of different sizes in ONE library call (gdt_ingest_resize_u8_batch: three
    launches for the whole list when the images are RGB).  ``plans``: per image ``((out_w, out_h), (fx, fy), box or None)``.
    Returns (list of uint8 outputs or None, list of fp32 C x h x w outputs or None)."""
    lib = _hip.load()
    if not images:
        return ([] if want_u8 else None), ([] if want_chw else None)
    dev, c = images[0].device, images[0].shape[2]
    images = [img.contiguous() for img in images]
    for img in images:
        if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != c or img.device != dev:
            raise ValueError("ingest needs uint8 H x W x C tensors with the same channel count on one HIP device")
    n = len(images)
    items = (_hip.IngestItem * n)()
    u8 = [None] * n
    chw = [None] * n
    # one allocation per output kind for the whole list (views of 16-byte aligned slices): 64 small allocations cost more than the kernels
    counts = [c * size[0] * size[1] for size, _, _ in plans]
    starts = [0] * n
    for i in range(1, n):
        starts[i] = starts[i - 1] + (counts[i - 1] + 15) // 16 * 16
    total = starts[-1] + counts[-1]
    flat_u8 = torch.empty(total, dtype=torch.uint8, device=dev) if want_u8 else None
    flat_f = torch.empty(total, dtype=torch.float32, device=dev) if want_chw else None
    for i, (img, (size, factors, box)) in enumerate(zip(images, plans)):
        it = items[i]
        it.src, it.h, it.w = img.data_ptr(), img.shape[0], img.shape[1]
        it.fx, it.fy = factors
        if box is not None:
            it.box = (ctypes.c_float * 4)(*[float(v) for v in box])
        it.out_w, it.out_h = size
        if want_u8:
            u8[i] = flat_u8[starts[i]:starts[i] + counts[i]].view(size[1], size[0], c)
            it.dst_hwc = u8[i].data_ptr()
        if want_chw:
            chw[i] = flat_f[starts[i]:starts[i] + counts[i]].view(c, size[1], size[0])
            it.dst_chw = chw[i].data_ptr()
    mean, std = mean_std if mean_std is not None else (None, None)
    need = ctypes.c_size_t()
    with torch.cuda.device(dev):
        _hip.check(lib.gdt_ingest_batch_workspace_bytes(items, n, c, ctypes.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        _hip.check(lib.gdt_ingest_resize_u8_batch(items, n, c, _f(mean, c), _f(std, c), ws.data_ptr(), ws.numel(),
                                                  torch.cuda.current_stream(dev).cuda_stream))
    return (u8 if want_u8 else None), (chw if want_chw else None)


def ingest_many(images, imsize, mean, std, clahe_clip=None, clahe_grid=8, streams=4):
    """``ingest`` over a list of decoded images of different sizes (the reference is batch-1 for exactly that reason,
    imageretrievalnet.py:319-322): the resampling of the whole list is one library call / three launches (``resize_many``); only the
    optional CLAHE still runs per image (its tile grid depends on the image size).  ``streams`` is kept for callers of the earlier
    per-image form and ignored.  Returns a list of fp32 3 x h x w tensors in input order."""
    if not images:
        return []
    sizes = list(imsize) if isinstance(imsize, (list, tuple)) else [imsize] * len(images)      # (per image: the dataset scales cropped queries, genericdataset.py:89-91)
    plans = [_plan(img, s) for img, s in zip(images, sizes)]
    if clahe_clip is None:
        return resize_many(images, plans, want_u8=False, mean_std=(mean, std), want_chw=True)[1]
    from . import clahe
    unit = resize_many(images, plans, want_u8=False, want_chw=True)[1]                              # [0, 1] RGB planes
    return [clahe.clahe_lab(u[None], clahe_clip, clahe_grid, None, (mean, std))[0] for u in unit]


class DeviceTransform:
    """Device-side counterpart of the ``.transform`` the hub attaches to a network (mdir/hub/model.py:38-42 builds it from
    ``runtime.data.transforms``, e.g. ``pil2np | apply_clahe:1.0 | totensor | normalize``): takes the DECODED image as a uint8
    H x W x 3 tensor on the device (optionally the ``imsize`` of the dataset's ``imresize``) and returns the fp32 3 x h x w tensor
    the network expects.  Supported steps: pil2np, apply_clahe[:clip[:grid[:lab]]], totensor, normalize."""

    def __init__(self, augmentations, mean_std):
        self.mean, self.std = [float(v) for v in mean_std[0]], [float(v) for v in mean_std[1]]
        self.clahe_clip, self.clahe_grid, self.normalize = None, 8, False
        self.spec = augmentations
        for step in [t.strip() for t in augmentations.split("|") if t.strip()]:
            name, *args = step.split(":")
            if name in ("pil2np", "totensor"):
                continue
            if name == "normalize":
                self.normalize = True
            elif name == "apply_clahe":
                self.clahe_clip = float(args[0]) if args else 4.0          # ApplyClahe defaults (photometric_transforms.py:31)
                self.clahe_grid = int(args[1]) if len(args) > 1 else 8
                if len(args) > 2 and args[2].lower() != "lab":
                    raise NotImplementedError("Colorspace %s is not supported on the HIP path" % args[2])
            else:
                raise KeyError("transform '%s' has no device implementation" % name)

    def __call__(self, img, imsize=None):
        mean, std = (self.mean, self.std) if self.normalize else ([0.0] * 3, [1.0] * 3)
        return ingest(img, imsize, mean, std, self.clahe_clip, self.clahe_grid)

    def __repr__(self):
        return "%s(%s, mean=%s, std=%s)" % (type(self).__name__, self.spec, self.mean, self.std)
