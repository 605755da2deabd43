"""CLAHE on the device -- the step between generator and embedder in the augment -> embed chain.

Reference (per image, on the CPU through cv2): ``ClahePost.postprocess`` (mdir/components/data/wrapper.py:325-348) ->
``ImageClahe.apply`` (mdir/components/data/transform/functional.py:151-158) -> ``apply_lightness_transform`` (:81-85) with
``rgb2normspace`` / ``normspace2rgb`` (:28-36 / :55-63, colorspace "lab") and ``ChannelClahe.apply_clahe`` (:147-148).
Here the whole batch is processed by three HIP launches (gandtr_amd/csrc/clahe.hip); there is no CPU fallback."""
import ctypes

import torch

from . import _hip


def _workspace(lib, n, h, w, tiles_x, tiles_y, device):
    need = ctypes.c_size_t()
    _hip.check(lib.gdt_clahe_workspace_bytes(n, h, w, tiles_x, tiles_y, ctypes.byref(need)))
    return torch.empty(need.value, dtype=torch.uint8, device=device)


def _float3(v):
    if v is None:
        return None
    flat = isinstance(v, (list, tuple)) and all(isinstance(t, (int, float)) for t in v)
    vals = [float(t) for t in (v if flat else torch.as_tensor(v).reshape(-1).tolist())]
    if len(vals) == 1:
        vals = vals * 3
    if len(vals) != 3:
        raise ValueError("expected 3 per-channel values, got %d" % len(vals))
    return (ctypes.c_float * 3)(*vals)


def clahe_u8(planes, clip_limit, grid_size=8):
    """``cv2.createCLAHE(clip_limit, (grid, grid)).apply`` on a batch of uint8 planes [N][H][W] (or one [H][W]) on the device."""
    lib = _hip.load()
    if not planes.is_cuda or planes.dtype != torch.uint8:
        raise ValueError("clahe_u8 needs a uint8 tensor on a HIP device")
    squeeze = planes.dim() == 2
    src = (planes[None] if squeeze else planes).contiguous()
    if src.dim() != 3:
        raise ValueError("Unsupported tensor dims: %s" % planes.dim())
    gx, gy = (grid_size, grid_size) if isinstance(grid_size, int) else grid_size
    n, h, w = src.shape
    dst = torch.empty_like(src)
    with torch.cuda.device(src.device):
        ws = _workspace(lib, n, h, w, gx, gy, src.device)
        _hip.check(lib.gdt_clahe_u8(src.data_ptr(), dst.data_ptr(), n, h, w, float(clip_limit), gx, gy, ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream(src.device).cuda_stream))
    return dst[0] if squeeze else dst


def clahe_lab(x, clip_limit, grid_size=8, in_meanstd=None, out_meanstd=None):
    """ImageClahe(clip_limit, grid_size, "lab") on a batch of RGB images N x 3 x H x W (fp32, device).  ``in_meanstd`` =
    (mean, std) un-normalises the input first (rgb = x * std + mean), ``out_meanstd`` normalises the result ((rgb - mean) / std);
    ClahePost uses the same pair for both."""
    lib = _hip.load()
    if not x.is_cuda:
        raise ValueError("clahe_lab needs a tensor on a HIP device")
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("clahe_lab expects N x 3 x H x W, got %s" % (tuple(x.shape),))
    x = x.detach().contiguous().float()
    n, _, h, w = x.shape
    y = torch.empty_like(x)
    in_mean, in_std = in_meanstd if in_meanstd is not None else (None, None)
    out_mean, out_std = out_meanstd if out_meanstd is not None else (None, None)
    with torch.cuda.device(x.device):
        ws = _workspace(lib, n, h, w, grid_size, grid_size, x.device)
        _hip.check(lib.gdt_clahe_lab_f32(x.data_ptr(), y.data_ptr(), n, h, w, _float3(in_std), _float3(in_mean), _float3(out_mean),
                                         _float3(out_std), float(clip_limit), grid_size, grid_size, ws.data_ptr(), ws.numel(),
                                         torch.cuda.current_stream(x.device).cuda_stream))
    return y
