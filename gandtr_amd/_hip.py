"""ctypes binding of libgandtr_hip.so (C ABI: include/gandtr_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``make -C gandtr_amd/csrc``.  There is NO
fallback: if the library is missing the HIP path raises -- a CUDA/HIP device request never silently runs torch ops.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

_LIB_PATH = os.environ.get("GANDTR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgandtr_hip.so")
_lib = None

GDT_OK, GDT_ERR_INVALID, GDT_ERR_HIP, GDT_ERR_WORKSPACE, GDT_ERR_NOT_CONVERGED = 0, 1, 2, 3, 4


class ConvDesc(ctypes.Structure):
    """struct gdt_conv_desc (include/gandtr_hip.h)."""
    _fields_ = [("cin", c_int), ("cout", c_int), ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int),
                ("pad_reflect", c_int), ("transposed", c_int), ("relu", c_int), ("out_f32_nchw", c_int),
                ("act", c_int), ("bn_eps", c_float)]


class IngestItem(ctypes.Structure):
    """struct gdt_ingest_item (include/gandtr_hip.h)."""
    _fields_ = [("src", c_void_p), ("h", c_int), ("w", c_int), ("fx", c_int), ("fy", c_int), ("box", c_float * 4),
                ("out_w", c_int), ("out_h", c_int), ("dst_hwc", c_void_p), ("dst_chw", c_void_p)]


class JpegInfo(ctypes.Structure):
    """struct gdt_jpeg_info (include/gandtr_hip.h)."""
    _fields_ = [("width", c_int), ("height", c_int), ("ncomp", c_int), ("hs", c_int * 4), ("vs", c_int * 4), ("tq", c_int * 4),
                ("td", c_int * 4), ("ta", c_int * 4), ("restart_interval", c_int), ("mcus_x", c_int), ("mcus_y", c_int),
                ("blocks_per_mcu", c_int), ("nsegments", c_int), ("scan_offset", ctypes.c_ulonglong),
                ("scan_capacity", ctypes.c_ulonglong), ("quant", (ctypes.c_ushort * 64) * 4), ("huff_bits", (ctypes.c_ubyte * 17) * 4),
                ("huff_vals", (ctypes.c_ubyte * 256) * 4), ("progressive", c_int), ("comp_id", c_int * 4), ("adobe_transform", c_int)]


class Level(ctypes.Structure):
    """struct gdt_level (include/gandtr_hip.h): one geometry of gdt_net_forward_levels."""
    _fields_ = [("x", c_void_p), ("n", c_int), ("h", c_int), ("w", c_int), ("rh", c_int), ("rw", c_int), ("rscale", c_float),
                ("outputs", POINTER(c_void_p)), ("n_outputs", c_int), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class JpegItem(ctypes.Structure):
    """struct gdt_jpeg_item (include/gandtr_hip.h)."""
    _fields_ = [("info", POINTER(JpegInfo)), ("scan", c_void_p), ("seg_off", POINTER(ctypes.c_uint)), ("dst_hwc", c_void_p)]


_FP = POINTER(c_float)
_IP = POINTER(c_int)

# name -> (restype, argtypes); every symbol declared in include/gandtr_hip.h
SIGNATURES = {
    "gdt_last_error": (c_char_p, []),
    "gdt_version": (c_char_p, []),
    "gdt_net_create": (c_int, [POINTER(c_void_p)]),
    "gdt_net_destroy": (None, [c_void_p]),
    "gdt_net_set_precision": (c_int, [c_void_p, c_int]),
    "gdt_net_input": (c_int, [c_void_p, c_int, _IP, _FP, _FP, _IP]),
    "gdt_net_conv": (c_int, [c_void_p, c_int, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_int, _IP]),
    "gdt_net_instance_norm": (c_int, [c_void_p, c_int, c_float, c_int, c_int, _IP]),
    "gdt_net_maxpool": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _IP]),
    "gdt_net_gem_l2n": (c_int, [c_void_p, c_int, c_float, c_float, c_float, _IP]),
    "gdt_net_output_nchw": (c_int, [c_void_p, c_int, c_void_p, _IP]),
    "gdt_net_hed_head": (c_int, [c_void_p, _IP, POINTER(c_void_p), _FP, _FP, c_float, c_int, _IP]),
    "gdt_net_finalize": (c_int, [c_void_p]),
    "gdt_net_output_shape": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _IP, _IP]),
    "gdt_net_num_outputs": (c_int, [c_void_p]),
    "gdt_net_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_size_t)]),
    "gdt_net_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, POINTER(c_void_p),
                                c_int, c_void_p, c_size_t, c_void_p]),
    "gdt_net_forward_levels": (c_int, [c_void_p, POINTER(Level), c_int, c_void_p]),
    "gdt_net_levels_joined": (c_int, [c_void_p, _IP]),
    "gdt_net_set_group_factor": (c_int, [c_void_p, c_float]),
    "gdt_net_flops": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_double)]),
    "gdt_net_set_profiling": (c_int, [c_void_p, c_int]),
    "gdt_net_profile_read": (c_int, [c_void_p, c_int, _IP, _IP, _IP, POINTER(c_double), POINTER(c_double)]),
    "gdt_net_profile_read_bytes": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_double)]),
    "gdt_net_num_ops": (c_int, [c_void_p]),
    "gdt_net_plan_summary": (c_int, [c_void_p, c_int, c_int, c_int, c_int, _IP, c_int]),
    "gdt_ms_aggregate": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gdt_whiten": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gdt_whiten_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gdt_retrieval_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "gdt_retrieval_scores_ranks": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                           c_void_p]),
    "gdt_retrieval_select_negatives": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                               c_int, c_int, c_void_p]),
    "gdt_l2n_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "gdt_gem_l2n": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "gdt_mfma_only_tflops": (c_int, [c_int, POINTER(c_double), c_void_p]),
    "gdt_ingest_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "gdt_ingest_resize_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_int, c_void_p, c_void_p,
                                     POINTER(c_float), POINTER(c_float), c_void_p, c_size_t, c_void_p]),
    "gdt_ingest_batch_workspace_bytes": (c_int, [POINTER(IngestItem), c_int, c_int, POINTER(c_size_t)]),
    "gdt_ingest_resize_u8_batch": (c_int, [POINTER(IngestItem), c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p, c_size_t,
                                           c_void_p]),
    "gdt_jpeg_parse": (c_int, [c_void_p, c_size_t, POINTER(JpegInfo)]),
    "gdt_jpeg_extract_scan": (c_int, [c_void_p, c_size_t, POINTER(JpegInfo), c_void_p, POINTER(ctypes.c_uint)]),
    "gdt_jpeg_parse_batch": (c_int, [POINTER(c_void_p), POINTER(c_size_t), c_int, POINTER(JpegInfo), _IP, c_int]),
    "gdt_jpeg_extract_scan_batch": (c_int, [POINTER(c_void_p), POINTER(c_size_t), POINTER(JpegInfo), c_int, c_void_p, POINTER(c_size_t),
                                            POINTER(ctypes.c_uint), POINTER(c_size_t), c_int]),
    "gdt_jpeg_decode_workspace_bytes": (c_int, [POINTER(JpegItem), c_int, POINTER(c_size_t)]),
    "gdt_jpeg_decode_u8_batch": (c_int, [POINTER(JpegItem), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gdt_jpeg_progressive_coefficients": (c_int, [c_char_p, c_size_t, POINTER(JpegInfo), c_void_p]),
    "gdt_jpeg_progressive_coefficients_batch": (c_int, [POINTER(c_void_p), POINTER(c_size_t), POINTER(JpegInfo), c_int, c_void_p, POINTER(c_size_t),
                                                        POINTER(c_int), c_int]),
    "gdt_jpeg_decode_coef_workspace_bytes": (c_int, [POINTER(JpegInfo), c_int, POINTER(c_size_t)]),
    "gdt_jpeg_decode_coef_u8_batch": (c_int, [POINTER(JpegInfo), c_void_p, POINTER(c_size_t), POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]),
    "gdt_whiten_learn_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "gdt_whiten_learn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int),
                                 c_void_p, c_size_t, c_void_p]),
    "gdt_clahe_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "gdt_clahe_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gdt_clahe_lab_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), POINTER(c_float),
                                  POINTER(c_float), c_double, c_int, c_int, c_void_p, c_size_t, c_void_p]),
}


class HipLibraryMissing(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load (once) and return the ctypes handle; raises HipLibraryMissing when the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: the gandtr HIP path has no CPU fallback. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C gandtr_amd/csrc`." % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here == ABI mismatch between header and library
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Translate a gdt status into the exception type the reference raises for the same condition
    (ValueError for bad arguments / unsupported shapes, RuntimeError for device failures)."""
    if rc == GDT_OK:
        return
    msg = load().gdt_last_error().decode("utf8", "replace")
    if rc == GDT_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError("gandtr_hip error %d: %s" % (rc, msg))
