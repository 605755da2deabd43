"""Host-side driver of the HIP graph executor (C ABI: include/gandtr_hip.h, binding: gandtr_amd/_hip.py).

``HipNet`` mirrors the builder calls one-to-one; the ``build_*`` functions translate the reference's state dicts
(names as produced by the reference modules, see gandtr_amd/tools/synth.py) into layer graphs:

  build_generator   ResnetGenerator            mdir/components/model/network/p2p_networks.py:269-313, :454-506
  build_embedder    ImageRetrievalNet (GeM)    mdir/external/cirtorch/networks/imageretrievalnet.py:101-123, :185-190
  build_hed         HedInterpolation           mdir/components/model/network/hed.py:30-83

PyTorch is used for device memory and streams only; no torch op runs on the data path.
"""
import collections
import ctypes
import math
import os

import numpy as np
import torch

from . import _hip
from ._hip import ConvDesc


def _f32(t):
    """host fp32 contiguous numpy view of a tensor / array (None passes through)"""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        t = t.detach().to("cpu", torch.float32).contiguous().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_SIDE_STREAMS = {}


def side_stream(device, k):
    """The k-th side stream of ``device``, shared by every net of the process.  torch hands out streams from a pool of 32 per device round-robin and the ROCm runtime
    multiplexes them onto a few hardware queues (four by default): the first net of a process got pool streams that sit on queues of their own, the third net's
    landed on the queue of the DEFAULT stream -- its pyramid levels then queued behind each other and every launch cost the host twice as much (measured: the
    same batch-1 multi-scale loop at 360 descriptors/s in a fresh process and 205 after two other networks had run in it).  One list per device keeps every net
    on the same first streams."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) <= k:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[k]


class HipNet:
    """One layer graph living on one GPU.  Host calls are serialised per handle (one thread at a time; see forward_many for what may overlap on the device)."""

    PRECISIONS = {"f16": 0, "f16x3": 1, "f16c": 2, "f16ch": 3}       # include/gandtr_hip.h, gdt_net_set_precision

    def __init__(self, device, precision="f16"):
        self.lib = _hip.load()
        if precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        self.precision = precision
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("HipNet needs a cuda (HIP) device, got %s" % self.device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.handle = ctypes.c_void_p()
        _hip.check(self.lib.gdt_net_create(ctypes.byref(self.handle)))
        _hip.check(self.lib.gdt_net_set_precision(self.handle, self.PRECISIONS[precision]))
        self.in_channels = None
        self._ws = None
        self._finalized = False
        # hipGraph replay of whole forwards (launch-bound small batches: the reference's own operating point is batch 1)
        self.use_graphs = os.environ.get("GANDTR_HIP_GRAPHS", "0") == "1"   # measured: no gain (kernels, not launches, bound batch 1)
        self.graph_max_workspace = 2 << 30      # geometries needing more scratch than this run eagerly
        self._graphs = collections.OrderedDict()   # key -> dict(graph, x, outs, ws) ; LRU of 8
        self._seen = {}

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value and self.lib is not None:
            try:
                self.lib.gdt_net_destroy(h)
            except Exception:       # interpreter shutdown: ctypes may already be torn down
                pass
            self.handle = None

    # ---- builder -------------------------------------------------------------------------------------------------
    def input(self, channels, perm=None, scale=None, shift=None):
        out = ctypes.c_int()
        perm_a = (ctypes.c_int * channels)(*perm) if perm is not None else None
        scale_a = (ctypes.c_float * channels)(*scale) if scale is not None else None
        shift_a = (ctypes.c_float * channels)(*shift) if shift is not None else None
        _hip.check(self.lib.gdt_net_input(self.handle, channels, perm_a, scale_a, shift_a, ctypes.byref(out)))
        self.in_channels = channels
        return out.value

    def conv(self, x, weight, bias=None, bn=None, stride=1, pad=0, reflect=False, transposed=False, relu=False,
             residual=-1, out_f32=False, act=0):
        w = _f32(weight)
        cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        d = ConvDesc(cin, cout, w.shape[2], w.shape[3], stride, pad, int(reflect), int(transposed), int(relu),
                     int(out_f32), act, 1e-5)
        b = _f32(bias)
        g = be = m = v = None
        if bn is not None:
            g, be, m, v = (_f32(t) for t in bn)
        out = ctypes.c_int()
        _hip.check(self.lib.gdt_net_conv(self.handle, x, ctypes.byref(d), _ptr(w), _ptr(b), _ptr(g), _ptr(be), _ptr(m),
                                         _ptr(v), residual, ctypes.byref(out)))
        return out.value

    def instance_norm(self, x, relu=False, residual=-1, eps=1e-5):
        out = ctypes.c_int()
        _hip.check(self.lib.gdt_net_instance_norm(self.handle, x, eps, int(relu), residual, ctypes.byref(out)))
        return out.value

    def maxpool(self, x, kernel, stride, pad=0):
        out = ctypes.c_int()
        _hip.check(self.lib.gdt_net_maxpool(self.handle, x, kernel, stride, pad, ctypes.byref(out)))
        return out.value

    def gem_l2n(self, x, p, eps_gem=1e-6, eps_l2=1e-6):
        out = ctypes.c_int()
        _hip.check(self.lib.gdt_net_gem_l2n(self.handle, x, float(p), eps_gem, eps_l2, ctypes.byref(out)))
        return out.value

    def output_nchw(self, x, bias=None):
        out = ctypes.c_int()
        b = _f32(bias)
        _hip.check(self.lib.gdt_net_output_nchw(self.handle, x, _ptr(b), ctypes.byref(out)))
        return out.value

    def hed_head(self, feats, score_w, score_b, fusion_w, fusion_b, sigmoid=True):
        ws = [_f32(w).reshape(-1) for w in score_w]
        wp = (ctypes.c_void_p * 5)(*[w.ctypes.data for w in ws])
        out = ctypes.c_int()
        _hip.check(self.lib.gdt_net_hed_head(self.handle, (ctypes.c_int * 5)(*feats), wp,
                                             (ctypes.c_float * 5)(*[float(b) for b in score_b]),
                                             (ctypes.c_float * 5)(*[float(w) for w in fusion_w]), float(fusion_b),
                                             int(sigmoid), ctypes.byref(out)))
        return out.value

    def finalize(self):
        with torch.cuda.device(self.device):
            _hip.check(self.lib.gdt_net_finalize(self.handle))
        self._finalized = True
        return self

    # ---- execution -----------------------------------------------------------------------------------------------
    @staticmethod
    def resized_size(h, w, scale):
        """Output size of F.interpolate(scale_factor=s): floor(float(in * s)) (torch/nn/functional.py)."""
        if scale is None:
            return h, w
        return int(math.floor(float(h * scale))), int(math.floor(float(w * scale)))

    #: knobs the planner reads when it plans a geometry (A/B inside one process): part of the key of the per-geometry cache below
    _PLAN_KNOBS = ("GDT_CONV_XEXP", "GDT_XEXP_CHAIN", "GDT_CONV_BNECK", "GDT_CONV_HALO_X3", "GDT_CONV_HALO_X3_FORMS", "GDT_X3_NORM_FOLD")

    def _geometry(self, n, rh, rw):
        """(workspace bytes, output shapes) of a geometry, planned once: every query plans the whole graph (make_plan, csrc/net.hip: ~0.1 ms for ResNet-101), and a
        forward asks three times per pyramid level -- 1.2 ms of the 8.6 ms a synchronised multi-scale call took (round 5)."""
        key = (n, rh, rw, getattr(self, "_group_factor", 1.0)) + tuple(os.environ.get(k) for k in self._PLAN_KNOBS)
        cache = self.__dict__.setdefault("_geo_cache", {})
        hit = cache.get(key)
        if hit is None:
            b = ctypes.c_size_t()
            _hip.check(self.lib.gdt_net_workspace_bytes(self.handle, n, rh, rw, ctypes.byref(b)))
            shapes = []
            dims, ndim = (ctypes.c_int * 4)(), ctypes.c_int()
            for slot in range(self.lib.gdt_net_num_outputs(self.handle)):
                _hip.check(self.lib.gdt_net_output_shape(self.handle, slot, n, rh, rw, dims, ctypes.byref(ndim)))
                shapes.append(tuple(dims[i] for i in range(ndim.value)))
            if len(cache) > 256:
                cache.clear()
            hit = (b.value, shapes)
            if self._finalized:
                cache[key] = hit
        return hit

    def output_shapes(self, n, rh, rw):
        return list(self._geometry(n, rh, rw)[1])

    PLAN_KEYS = ("conv_launches", "bottlenecks_fused", "conv3x3_expand", "chained_reduce", "shortcuts_folded", "norms_folded", "pools_fused", "direct_stem",
                 "transposed_fused", "stride2_shift")

    def plan_summary(self, n, rh, rw, resize=False):
        """the planner's fusion decisions for a geometry as a dict of counts (gdt_net_plan_summary: host logic, no device call)"""
        c = (ctypes.c_int * 10)()
        _hip.check(self.lib.gdt_net_plan_summary(self.handle, n, rh, rw, int(bool(resize)), c, 10))
        return dict(zip(self.PLAN_KEYS, list(c)))

    def flops(self, n, rh, rw):
        f = ctypes.c_double()
        _hip.check(self.lib.gdt_net_flops(self.handle, n, rh, rw, ctypes.byref(f)))
        return f.value

    def set_profiling(self, enable):
        self._profiling = bool(enable)          # profiled forwards run eagerly (events are recorded per op)
        _hip.check(self.lib.gdt_net_set_profiling(self.handle, int(enable)))

    def profile(self):
        """per-op (kind, conv N-tile, ms, algorithmic flops) of the last profiled forward"""
        cap = max(1, int(self.lib.gdt_net_num_ops(self.handle)))
        n = ctypes.c_int()
        kinds, tiles = (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
        ms, fl = (ctypes.c_double * cap)(), (ctypes.c_double * cap)()
        _hip.check(self.lib.gdt_net_profile_read(self.handle, cap, ctypes.byref(n), kinds, tiles, ms, fl))
        return [(kinds[i], tiles[i], ms[i], fl[i]) for i in range(n.value)]

    def profile_bytes(self):
        """per-op algorithmic HBM bytes of the last profiled forward (same op order as profile())"""
        cap = max(1, int(self.lib.gdt_net_num_ops(self.handle)))
        n = ctypes.c_int()
        by = (ctypes.c_double * cap)()
        _hip.check(self.lib.gdt_net_profile_read_bytes(self.handle, cap, ctypes.byref(n), by))
        return [by[i] for i in range(n.value)]

    def workspace_bytes(self, n, rh, rw):
        return self._geometry(n, rh, rw)[0]

    def set_group_factor(self, factor):
        """planner hint for the geometry planned next: it runs concurrently with others of this net; factor = (pixels of all of them) / (its own), 1 = alone
        (gdt_net_set_group_factor).  forward_many sets it per level and resets it."""
        factor = max(1.0, float(factor))
        if factor != getattr(self, "_group_factor", 1.0):
            _hip.check(self.lib.gdt_net_set_group_factor(self.handle, factor))
            self._group_factor = factor

    def _launch(self, x, n, h, w, rh, rw, rscale, ws, outs):
        optrs = (ctypes.c_void_p * max(1, len(outs)))(*[o.data_ptr() for o in outs])
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _hip.check(self.lib.gdt_net_forward(self.handle, x.data_ptr(), n, h, w, rh, rw, rscale, optrs, len(outs),
                                            ws.data_ptr(), ws.numel(), stream))

    def forward(self, x, scale=None):
        """x: fp32 NCHW tensor on this net's device.  ``scale``: optional F.interpolate scale_factor applied to the
        input inside the pack kernel.  Returns the list of external outputs (torch tensors on the device).

        A geometry seen for the second time is captured into a hipGraph (static input / output / scratch buffers) and
        replayed from then on: one graph launch instead of ~100 kernel launches, which is what bounds small batches."""
        if not self._finalized:
            raise RuntimeError("HipNet.forward before finalize()")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError("expected an N x %s x H x W input, got %s" % (self.in_channels, tuple(x.shape)))
        if x.device != self.device:
            x = x.to(self.device)
        x = x.contiguous().float()
        n, _, h, w = x.shape
        rh, rw = self.resized_size(h, w, scale)
        rscale = float(np.float32(1.0 / scale)) if scale is not None else 1.0
        key = (n, h, w, rh, rw, rscale)
        with torch.cuda.device(x.device):
            entry = self._graphs.get(key) if (self.use_graphs and not getattr(self, "_profiling", False)) else None      # (a profiled forward runs eagerly: events per op)
            if entry is not None:
                self._graphs.move_to_end(key)
                entry["x"].copy_(x)
                entry["graph"].replay()
                return [o.clone() for o in entry["outs"]]
            need = self.workspace_bytes(n, rh, rw)
            shapes = self.output_shapes(n, rh, rw)
            profiling = getattr(self, "_profiling", False)
            if self.use_graphs and not profiling and need <= self.graph_max_workspace and self._seen.get(key, 0) >= 1 \
                    and not torch.cuda.is_current_stream_capturing():
                entry = self._capture(x, key, need, shapes)
                if entry is not None:
                    return [o.clone() for o in entry["outs"]]
            self._seen[key] = self._seen.get(key, 0) + 1
            if len(self._seen) > 256:
                self._seen.clear()
            if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            outs = [torch.empty(s, dtype=torch.float32, device=x.device) for s in shapes]
            self._launch(x, n, h, w, rh, rw, rscale, self._ws, outs)
        return outs

    def forward_many(self, inputs):
        """Several independent forwards -- ``inputs`` = [(x, scale or None), ...], e.g. the levels of the multi-scale pyramid
        (CirMultiscaleAggregation, wrapper.py:225-263) -- issued on one side stream each, every one with its own scratch buffer, and joined
        on the caller's stream.  A level at batch 8 leaves most of the chip idle (layer3 of ResNet-101 at scale 1/2: 32 patch tiles for
        256 CUs); the levels' kernels fill each other's gaps.  Results are those of ``forward`` called level by level.

        Invariant this relies on (and the C handle states, include/gandtr_hip.h): the HOST calls on one ``gdt_net`` are serialised -- each
        ``gdt_net_forward`` plans its geometry into the handle and enqueues its launches before it returns -- and every byte of per-forward
        DEVICE state lives in the caller's workspace, so several forwards of one handle may be in flight on different streams as long as each
        has its own workspace and they are issued from one host thread at a time.  The side workspaces are kept between calls up to
        ``side_workspace_cap`` bytes in total (default 8 GiB); beyond it they are released when the call returns."""
        if not self._finalized:
            raise RuntimeError("HipNet.forward_many before finalize()")
        if len(inputs) == 1 or os.environ.get("GANDTR_HIP_CONCURRENT_LEVELS", "1") == "0":
            return [self.forward(x, scale=s) for x, s in inputs]
        if os.environ.get("GANDTR_HIP_JOINT_LEVELS", "0") == "1":           # (opt-in: measured 5-12 % slower than the side streams, csrc/gdt_common.h MultiConv)
            return self._forward_levels(inputs)
        if getattr(self, "_profiling", False):
            return [self.forward(x, scale=s) for x, s in inputs]
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        pools = self.__dict__.setdefault("_side", {"streams": [], "ws": []})
        while len(pools["streams"]) < len(inputs):
            pools["streams"].append(side_stream(dev, len(pools["streams"])))
            pools["ws"].append(None)
        group_px = float(sum(x.shape[0] * math.prod(self.resized_size(x.shape[2], x.shape[3], s)) for x, s in inputs if x.dim() == 4))
        try:
          with torch.cuda.device(dev):
            # GANDTR_HIP_FIRST_ON_CURRENT=1 (experiment, off): the first input on the CALLER's stream, issued last, the others on side streams issued first -- one
            # stream fewer, so that the default stream, two side streams and a collective's stream have a hardware queue each (the runtime has four).  Measured:
            # 8 x 1024^2 6.68 -> 6.80 ms, 1 x 1024^2 2.87 -> 2.78 ms, one sharded rank 1127 -> 1100 descriptors/s: no
            first_on_cur = os.environ.get("GANDTR_HIP_FIRST_ON_CURRENT", "0") == "1"
            results = [None] * len(inputs)
            order = (list(range(1, len(inputs))) + [0]) if first_on_cur else list(range(len(inputs)))
            # (the order in which the host enqueues the levels -- as given, smallest first, largest first -- makes no difference: hub scales at 8 x 1024^2 6.36 / 6.56 /
            # 6.43 ms, {1, 1/sqrt 2, sqrt 2} 13.68 / 13.64 / 13.67 ms)
            for k in order:
                x, scale = inputs[k]
                if x.dim() != 4 or x.shape[1] != self.in_channels:
                    raise ValueError("expected an N x %s x H x W input, got %s" % (self.in_channels, tuple(x.shape)))
                x = x.to(dev).contiguous().float()
                n, _, h, w = x.shape
                rh, rw = self.resized_size(h, w, scale)
                rscale = float(np.float32(1.0 / scale)) if scale is not None else 1.0
                self.set_group_factor(round(group_px / float(n * rh * rw), 3))       # the levels run together: fusion thresholds count the group's patches
                need = self.workspace_bytes(n, rh, rw)
                shapes = self.output_shapes(n, rh, rw)
                on_cur = first_on_cur and k == 0
                st = cur if on_cur else pools["streams"][k]
                if not on_cur:
                    st.wait_stream(cur)
                with torch.cuda.stream(st):
                    if pools["ws"][k] is None or pools["ws"][k].numel() < need:
                        pools["ws"][k] = None
                        pools["ws"][k] = torch.empty(need, dtype=torch.uint8, device=dev)
                    outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
                    self._launch(x, n, h, w, rh, rw, rscale, pools["ws"][k], outs)
                    if not on_cur:
                        x.record_stream(st)
                results[k] = outs
            for k in range(len(inputs)):
                if first_on_cur and k == 0:
                    continue
                cur.wait_stream(pools["streams"][k])
                for o in results[k]:
                    o.record_stream(cur)
            held = sum(w.numel() for w in pools["ws"] if w is not None)
            if held > getattr(self, "side_workspace_cap", 8 << 30):
                for k, w in enumerate(pools["ws"]):
                    if w is not None:
                        w.record_stream(cur)
                        pools["ws"][k] = None
        finally:
            self.set_group_factor(1.0)
        return results

    MAX_LEVELS = 4          # GDT_MAX_LEVELS (csrc/gdt_common.h): geometries per gdt_net_forward_levels call

    def levels_joined(self):
        """(ops whose levels shared one launch, launches the levels handed to the lock-step driver) of the last forward_many group"""
        n = ctypes.c_int()
        j = int(self.lib.gdt_net_levels_joined(self.handle, ctypes.byref(n)))
        return j, n.value

    def _forward_levels(self, inputs):
        """``inputs`` in groups of at most four geometries, each group ONE ``gdt_net_forward_levels`` call on the caller's stream: the ops run in lock-step and the
        levels' launches of an op are one launch wherever the kernel has a multi-geometry entry (1x1 convs, 3x3 patch convs, fused Bottlenecks; round 5).  OPT-IN
        (GANDTR_HIP_JOINT_LEVELS=1): 120 launches instead of 300 for a three-level ResNet-101 pyramid, but 5-12 % slower than one side stream per level
        (csrc/gdt_common.h, MultiConv: measurements).  Every level has its own scratch buffer (kept between calls up to ``side_workspace_cap`` bytes in total); results are those of
        ``forward`` level by level, bit for bit (tests/test_hip_f16c.py::test_forward_many_equals_level_by_level_forward)."""
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        pools = self.__dict__.setdefault("_side", {"streams": [], "ws": []})
        while len(pools["ws"]) < min(len(inputs), self.MAX_LEVELS):
            pools["streams"].append(side_stream(dev, len(pools["streams"])))
            pools["ws"].append(None)
        results = []
        with torch.cuda.device(dev):
            for lo in range(0, len(inputs), self.MAX_LEVELS):
                group = inputs[lo:lo + self.MAX_LEVELS]
                levels = (_hip.Level * len(group))()
                keep = []
                group_px = float(sum(x.shape[0] * math.prod(self.resized_size(x.shape[2], x.shape[3], s)) for x, s in group if x.dim() == 4))
                for k, (x, scale) in enumerate(group):
                    if x.dim() != 4 or x.shape[1] != self.in_channels:
                        raise ValueError("expected an N x %s x H x W input, got %s" % (self.in_channels, tuple(x.shape)))
                    x = x.to(dev).contiguous().float()
                    n, _, h, w = x.shape
                    rh, rw = self.resized_size(h, w, scale)
                    self.set_group_factor(group_px / float(n * rh * rw) if len(group) > 1 else 1.0)      # (gdt_net_forward_levels plans each level with the same factor)
                    need = self.workspace_bytes(n, rh, rw)
                    if pools["ws"][k] is None or pools["ws"][k].numel() < need:
                        pools["ws"][k] = None
                        pools["ws"][k] = torch.empty(need, dtype=torch.uint8, device=dev)
                    outs = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in self.output_shapes(n, rh, rw)]
                    optrs = (ctypes.c_void_p * max(1, len(outs)))(*[o.data_ptr() for o in outs])
                    lv = levels[k]
                    lv.x, lv.n, lv.h, lv.w, lv.rh, lv.rw = x.data_ptr(), n, h, w, rh, rw
                    lv.rscale = float(np.float32(1.0 / scale)) if scale is not None else 1.0
                    lv.outputs, lv.n_outputs = optrs, len(outs)
                    lv.workspace, lv.workspace_bytes = pools["ws"][k].data_ptr(), pools["ws"][k].numel()
                    keep.append((x, optrs))
                    results.append(outs)
                self.set_group_factor(1.0)
                _hip.check(self.lib.gdt_net_forward_levels(self.handle, levels, len(group), cur.cuda_stream))
                del keep
            held = sum(w.numel() for w in pools["ws"] if w is not None)
            if held > getattr(self, "side_workspace_cap", 8 << 30):
                for k, w in enumerate(pools["ws"]):
                    if w is not None:
                        w.record_stream(cur)
                        pools["ws"][k] = None
        return results

    def _capture(self, x, key, need, shapes):
        n, h, w, rh, rw, rscale = key
        try:
            sx = x.clone()
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            outs = [torch.empty(s, dtype=torch.float32, device=x.device) for s in shapes]
            side = side_stream(x.device, 0)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):           # warm-up outside capture (lazy function attributes etc.)
                self._launch(sx, n, h, w, rh, rw, rscale, ws, outs)
            torch.cuda.current_stream(x.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._launch(sx, n, h, w, rh, rw, rscale, ws, outs)
            g.replay()
        except Exception:               # capture is an optimisation only: fall back to eager launches for this geometry
            self._seen[key] = -(1 << 30)
            return None
        entry = {"graph": g, "x": sx, "outs": outs, "ws": ws}
        self._graphs[key] = entry
        while len(self._graphs) > 8:
            self._graphs.popitem(last=False)
        return entry

# ======================================================================================================= builders

def _bn(sd, p):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])


def generator_layout(sd):
    """(norm, ngf, n_blocks, in_nc, out_nc) recovered from state-dict keys: BatchNorm iff ``model.2.running_mean`` is
    present (use_bias rule p2p_networks.py:264-267)."""
    norm = "batch" if "model.2.running_mean" in sd else "instance"
    n_blocks = sum(1 for k in sd if k.endswith(".conv_block.1.weight"))
    w0 = sd["model.1.weight"]
    last = 10 + n_blocks + 6 + 1
    return norm, w0.shape[0], n_blocks, w0.shape[1], sd["model.%d.weight" % last].shape[0]


def build_generator(sd, device, taps=(), pre_tanh=False, in_affine=None, precision="f16c", finalize=True, norm=None):
    """ResnetGenerator as a HIP graph.  External outputs: [generator output] + one per requested tap (in ``taps``
    order).  Taps follow the reference's nn.Sequential indices (p2p_networks.py:316-334); a norm-layer tap aliases the
    post-ReLU tensor because the reference's ReLUs are in-place (:272).  Tap 0 / the second reflection pad are not
    materialised on the device (padding is resolved inside the conv loader) and are not available."""
    key_norm, ngf, n_blocks, in_nc, out_nc = generator_layout(sd)
    norm = norm or key_norm          # the module's configured norm type when the caller knows it (get_norm_layer, p2p_networks.py:23-35)
    if norm not in ("instance", "batch"):
        raise NotImplementedError('normalization layer [%s] is not found' % norm)          # p2p_networks.py:34
    if norm == "batch" and key_norm != "batch":
        raise NotImplementedError("BatchNorm without running statistics (track_running_stats=False) has no inference form on the HIP path")
    inorm = norm == "instance"
    net = HipNet(device, precision)
    tap_slots = {}

    def tap(idx, t, bias=None):
        if idx in taps and idx not in tap_slots:
            tap_slots[idx] = net.output_nchw(t, bias)

    def conv_norm_relu(x, key, nkey, idx, relu=True, residual=-1, **kw):
        """conv -> norm -> (ReLU) (+ residual).  InstanceNorm: bias-free conv (the bias cancels in the norm) + separate
        norm kernels; BatchNorm: folded into the conv epilogue."""
        if inorm:
            raw = net.conv(x, sd[key + ".weight"], None, **kw)
            if idx is not None:
                tap(idx, raw, sd.get(key + ".bias"))
            return net.instance_norm(raw, relu=relu, residual=residual)
        if idx is not None and idx in taps:   # raw conv output requested: unfused variant
            raw = net.conv(x, sd[key + ".weight"], sd.get(key + ".bias"), **kw)
            tap(idx, raw)
        return net.conv(x, sd[key + ".weight"], sd.get(key + ".bias"), bn=_bn(sd, nkey), relu=relu, residual=residual, **kw)

    if in_affine is not None:
        x = net.input(in_nc, scale=in_affine[0], shift=in_affine[1])
    else:
        x = net.input(in_nc)
    h = conv_norm_relu(x, "model.1", "model.2", 1, pad=3, reflect=True)
    tap(2, h); tap(3, h)
    i = 4
    for _ in range(2):
        h = conv_norm_relu(h, "model.%d" % i, "model.%d" % (i + 1), i, stride=2, pad=1)
        tap(i + 1, h); tap(i + 2, h)
        i += 3
    for _ in range(n_blocks):
        p = "model.%d.conv_block." % i
        r = conv_norm_relu(h, p + "1", p + "2", None, pad=1, reflect=True)
        h = conv_norm_relu(r, p + "5", p + "6", None, relu=False, residual=h, pad=1, reflect=True)
        tap(i, h)
        i += 1
    for _ in range(2):
        h = conv_norm_relu(h, "model.%d" % i, "model.%d" % (i + 1), i, transposed=True, stride=2, pad=1)
        tap(i + 1, h); tap(i + 2, h)
        i += 3
    head = "model.%d" % (i + 1)
    if (i + 1) in taps and not pre_tanh:
        tap_slots[i + 1] = net.conv(h, sd[head + ".weight"], sd[head + ".bias"], pad=3, reflect=True, out_f32=True, act=0)
    out = net.conv(h, sd[head + ".weight"], sd[head + ".bias"], pad=3, reflect=True, out_f32=True, act=0 if pre_tanh else 1)
    if (i + 2) in taps:
        tap_slots[i + 2] = out
    if finalize:
        net.finalize()
    net.out_slot = out
    net.tap_slots = tap_slots
    return net


VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
RESNET101_BLOCKS = (3, 4, 23, 3)


def _vgg16_trunk(net, x, sd, prefix="features."):
    i = 0
    for v in VGG16_CFG:
        if v == "M":
            x = net.maxpool(x, 2, 2)
            i += 1
        else:
            x = net.conv(x, sd["%s%d.weight" % (prefix, i)], sd["%s%d.bias" % (prefix, i)], pad=1, relu=True)
            i += 2
    return x


def _resnet_trunk(net, x, sd, prefix="features."):
    blocks = []
    for li in range(4):
        nb = 0
        while "%s%d.%d.conv1.weight" % (prefix, 4 + li, nb) in sd:
            nb += 1
        blocks.append(nb)
    x = net.conv(x, sd[prefix + "0.weight"], None, bn=_bn(sd, prefix + "1"), stride=2, pad=3, relu=True)
    x = net.maxpool(x, 3, 2, 1)
    for li, nb in enumerate(blocks):
        for b in range(nb):
            p = "%s%d.%d." % (prefix, 4 + li, b)
            stride = 2 if (b == 0 and li > 0) else 1
            idt = x
            if b == 0:
                idt = net.conv(x, sd[p + "downsample.0.weight"], None, bn=_bn(sd, p + "downsample.1"), stride=stride)
            o = net.conv(x, sd[p + "conv1.weight"], None, bn=_bn(sd, p + "bn1"), relu=True)
            o = net.conv(o, sd[p + "conv2.weight"], None, bn=_bn(sd, p + "bn2"), stride=stride, pad=1, relu=True)
            x = net.conv(o, sd[p + "conv3.weight"], None, bn=_bn(sd, p + "bn3"), relu=True, residual=idt)
    return x


def embedder_arch(sd):
    """'vgg16' or 'resnet101'-style trunk, recognised from the state-dict keys (imageretrievalnet.py:185-190)."""
    return "resnet" if "features.4.0.conv1.weight" in sd else "vgg16"


def build_embedder(sd, device, in_affine=None, feature_tap=False, precision="f16", finalize=True):
    """GeM embedder (ImageRetrievalNet.forward with lwhiten=None, whiten=None).  External output 0: descriptors as a
    row-major [N][D] fp32 matrix (the reference returns its transpose view, D x N)."""
    net = HipNet(device, precision)
    x = net.input(3, scale=in_affine[0], shift=in_affine[1]) if in_affine is not None else net.input(3)
    f = _resnet_trunk(net, x, sd) if embedder_arch(sd) == "resnet" else _vgg16_trunk(net, x, sd)
    net.out_slot = net.gem_l2n(f, float(sd["pool.p"].reshape(-1)[0]))
    net.feature_slot = net.output_nchw(f) if feature_tap else None
    if finalize:
        net.finalize()
    return net


HED_BLOCKS = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))


def build_hed(sd, device, perm=None, in_affine=None, sigmoid=True, precision="f16", finalize=True):
    """HedInterpolation.forward (hed.py:60-83); ``perm``/``in_affine`` fold the RgbToBgrPre + MeanStdPre wrappers
    (wrapper.py:351-364, :182-194) into the input pack kernel."""
    net = HipNet(device, precision)
    x = net.input(3, perm=perm, scale=None if in_affine is None else in_affine[0],
                  shift=None if in_affine is None else in_affine[1])
    feats = []
    for bi in range(5):
        off = 0
        if bi > 0:
            x = net.maxpool(x, 2, 2)
            off = 1
        ci = 0
        while "vgg%d.%d.weight" % (bi + 1, off + 2 * ci) in sd:
            k = "vgg%d.%d" % (bi + 1, off + 2 * ci)
            x = net.conv(x, sd[k + ".weight"], sd[k + ".bias"], pad=1, relu=True)
            ci += 1
        feats.append(x)
    net.out_slot = net.hed_head(
        feats, [sd["score%d.weight" % (k + 1)] for k in range(5)], [float(sd["score%d.bias" % (k + 1)]) for k in range(5)],
        [float(v) for v in sd["fusion.0.weight"].reshape(-1)], float(sd["fusion.0.bias"]), sigmoid)
    if finalize:
        net.finalize()
    return net


# ======================================================================================== stand-alone descriptor ops

def ms_aggregate(x, msp):
    """x: [S][N][D] fp32 cuda -> [N][D]  (wrapper.py:236-245, batched)."""
    lib = _hip.load()
    x = x.contiguous().float()
    s, n, d = x.shape
    y = torch.empty((n, d), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _hip.check(lib.gdt_ms_aggregate(x.data_ptr(), y.data_ptr(), s, n, d, float(msp),
                                        torch.cuda.current_stream(x.device).cuda_stream))
    return y


def whiten(v, P, m, dims=None, float64=False):
    """v: [N][D], P: [D][D], m: [D] or [D][1] (cuda) -> [N][dims]  (wrapper.py:320-322, batched).  ``float64``: the arithmetic of the
    reference's ``whiten`` stage (numpy float64, mdir/stages/whiten.py:20-23) instead of the float32 of the inference wrapper."""
    lib = _hip.load()
    dt = torch.float64 if float64 else torch.float32
    v = v.contiguous().to(dt)
    P = P.contiguous().to(dt)
    m = m.contiguous().to(dt).reshape(-1)
    n, d = v.shape
    dims = int(dims or P.shape[0])
    tmp = torch.empty((n, dims), dtype=dt, device=v.device)
    out = torch.empty((n, dims), dtype=dt, device=v.device)
    with torch.cuda.device(v.device):
        if float64:
            _hip.check(lib.gdt_whiten_f64(P.data_ptr(), m.data_ptr(), v.data_ptr(), tmp.data_ptr(), out.data_ptr(), n, d, dims,
                                          torch.cuda.current_stream(v.device).cuda_stream))
            return out
        _hip.check(lib.gdt_whiten(P.data_ptr(), m.data_ptr(), v.data_ptr(), tmp.data_ptr(), out.data_ptr(), n, d, dims,
                                  torch.cuda.current_stream(v.device).cuda_stream))
    return out


def gem_l2n(fmap, p=3.0, eps_gem=1e-6, eps_l2=1e-6):
    """fmap: N x D x h x w fp32 cuda (reference layout).  Returns (gem N x D x 1 x 1, l2n(gem) N x D x 1 x 1):
    cirtorch layers/functional.py:21-22, :130-131."""
    lib = _hip.load()
    if not fmap.is_cuda or fmap.dim() != 4:
        raise ValueError("gem_l2n needs an N x D x h x w tensor on a HIP device")
    fmap = fmap.contiguous().float()
    n, d, h, w = fmap.shape
    pooled = torch.empty((n, d), dtype=torch.float32, device=fmap.device)
    out = torch.empty((n, d), dtype=torch.float32, device=fmap.device)
    with torch.cuda.device(fmap.device):
        _hip.check(lib.gdt_gem_l2n(fmap.data_ptr(), n, d, h, w, float(p), eps_gem, eps_l2, pooled.data_ptr(), out.data_ptr(),
                                   torch.cuda.current_stream(fmap.device).cuda_stream))
    return pooled.view(n, d, 1, 1), out.view(n, d, 1, 1)


def l2n_rows(x, eps=1e-6):
    lib = _hip.load()
    x = x.contiguous().float()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _hip.check(lib.gdt_l2n_rows(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], eps,
                                    torch.cuda.current_stream(x.device).cuda_stream))
    return y
