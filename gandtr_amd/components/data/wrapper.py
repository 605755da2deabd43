"""Inference wrappers around ``model(x)`` -- host mirror of mdir/components/data/wrapper.py:
Compose :15-49, Wrapper :52-65, MeanStdPost/Pre :149-194, CirMultiscaleAggregation :197-263, FakeBatch :266-279,
CirFakeTupleBatch :282-305, CirtorchWhiten :308-322, ClahePost :325-348, RgbToBgrPre :351-364, registry :367-396.

Protocol (unchanged): ``preprocess(tensor, outputmodel) -> (tensor, meta)`` applied in list order,
``postprocess(tensor, outputmodel, meta) -> tensor`` applied in REVERSE order.

MI355X-specific behaviour (same results, different place of execution):
  * multi-scale: on a cuda model the pyramid is not materialised -- preprocess emits ``ScaledInput`` handles and the
    bilinear resize runs inside the HIP input-pack kernel; aggregation and whitening run as HIP kernels.
  * the reference's aggregate / whiten code only works for batch size 1; here a batch of N images
    yields the stack of the N per-image reference results (D x N, one column per image; N = 1 keeps the (D,) shape).
"""
import json

import numpy as np
import torch
import torch.nn.functional as F

from ...tools import tensors, utils
from ..model.network._hipbacked import ScaledInput


class Compose:
    def __init__(self, wrappers, device):
        self.wrappers = wrappers
        self.device = device

    def __call__(self, tensor, inference, outputmodel=None, tensor_params=None, fold_input=False):
        """``fold_input``: the caller guarantees that ``inference`` hands the wrapper-processed tensor straight to ``outputmodel``
        (SingleNetwork.forward does; a SequentialNetwork, whose hoisted wrappers act on the CHAIN input while ``outputmodel`` is the
        last member, never does) -- only then may trailing per-channel input wrappers run inside the model's input pack."""
        tensor_params = {} if tensor_params is None else tensor_params
        if not self.wrappers:
            if isinstance(tensor, torch.Tensor):
                tensor = tensor.to(self.device)
            return inference(tensor, **tensor_params)
        if outputmodel is None:
            outputmodel = inference
        active, folded = self._fold_input_wrappers(outputmodel) if fold_input else (self.wrappers, None)
        metas = []
        for w in active:
            tensor, meta = w.preprocess(tensor, outputmodel)
            metas.append(meta)
        if folded is None:
            tensor = inference(tensors.to_device(tensor, self.device), **tensor_params)
        else:                               # the trailing per-channel input wrappers run inside the model's HIP input-pack kernel
            tensor = inference(tensors.to_device(tensor, self.device), input_transform=folded, **tensor_params)
        for w, meta in reversed(list(zip(active, metas))):
            tensor = w.postprocess(tensor, outputmodel, meta)
        return tensor

    def _fold_input_wrappers(self, outputmodel):
        """(wrappers to run on the host, (perm, scale, shift) or None).  A model on a HIP device that declares
        ``accepts_input_transform`` takes the longest trailing run of pure per-channel input wrappers (channel permutation, mean / std
        re-normalisation: RgbToBgrPre, MeanStdPre -- mdir/components/data/wrapper.py:351-364, :182-194) as ONE transform
        ``y[c] = x[perm[c]] * scale[c] + shift[c]`` applied while its input is packed: no torch op on the data path."""
        if not getattr(outputmodel, "accepts_input_transform", False):
            return self.wrappers, None
        dev = getattr(outputmodel, "_hip_device", None)
        if dev is None or dev().type != "cuda":
            return self.wrappers, None
        nch = (getattr(outputmodel, "meta", None) or {}).get("in_channels")

        def foldable(w):
            tr = w.input_transform()
            return tr is not None and all(len(a) == nch for a in tr[1:])
        cut = len(self.wrappers)
        while cut > 0 and foldable(self.wrappers[cut - 1]):
            cut -= 1
        if cut == len(self.wrappers):
            return self.wrappers, None
        perm, scale, shift = None, None, None
        for w in self.wrappers[cut:]:
            kind, *args = w.input_transform()
            if perm is None:
                n = len(args[0])
                perm, scale, shift = list(range(n)), [1.0] * n, [0.0] * n
            if kind == "perm":
                p = args[0]
                perm, scale, shift = [perm[i] for i in p], [scale[i] for i in p], [shift[i] for i in p]
            else:
                s2, h2 = args
                scale, shift = [a * b for a, b in zip(scale, s2)], [a * b + c for a, b, c in zip(shift, s2, h2)]
        return self.wrappers[:cut], (tuple(perm), tuple(scale), tuple(shift))

    def __repr__(self):
        inner = "\n" + "".join("    %s\n" % w for w in self.wrappers) if self.wrappers else ""
        return "%s([%s])" % (type(self).__name__, inner)


class Wrapper:
    def __init__(self, device):
        pass

    def preprocess(self, tensor, _outputmodel):
        return tensor, None

    def postprocess(self, tensor, _outputmodel, _metadata):
        return tensor

    def input_transform(self):
        """("perm", [..]) or ("affine", scale, shift) when the wrapper is a pure per-channel transform of the model INPUT with an
        identity postprocess (then a HIP model can apply it while packing its input, Compose._fold_input_wrappers); else None."""
        return None


def _on_hip(t):
    return isinstance(t, torch.Tensor) and t.is_cuda


class MeanStdPost(Wrapper):
    """x * std_in + mean_in, then (. - mean_out) / std_out, on the network OUTPUT."""

    def __init__(self, input_meanstd, output_meanstd, device):
        super().__init__(device)
        input_meanstd, output_meanstd = json.loads(input_meanstd), json.loads(output_meanstd)
        if any(x == 0 for x in input_meanstd[1]) or any(x == 0 for x in output_meanstd[1]):
            raise ValueError("Some std element is zero, leading to zero division.")
        self.raw = (input_meanstd, output_meanstd)
        self.input_meanstd = [self.mean2tensor(x, device) for x in input_meanstd]
        self.output_meanstd = [self.mean2tensor(x, device) for x in output_meanstd]

    @staticmethod
    def mean2tensor(mean, device):
        mean = torch.as_tensor(mean, device=device)
        return mean[:, None, None] if mean.ndim == 1 else mean

    def affine(self):
        """Equivalent per-channel (scale, shift): y = x * scale + shift (used to fold the wrapper into a HIP input pack)."""
        (mi, si), (mo, so) = self.raw
        return [s / o for s, o in zip(si, so)], [(m - n) / o for m, n, o in zip(mi, mo, so)]

    def postprocess(self, tensor, outputmodel, meta):
        if isinstance(tensor, list):
            return [self.postprocess(x, outputmodel, meta) for x in tensor]
        return self._adapt(tensor)

    def _adapt(self, tensor):
        tensor = tensor.mul(self.input_meanstd[1]).add(self.input_meanstd[0])
        return tensor.sub(self.output_meanstd[0]).div(self.output_meanstd[1])

    def __repr__(self):
        return "%s(input_meanstd=%s,output_meanstd=%s)" % (type(self).__name__, self.input_meanstd, self.output_meanstd)


class MeanStdPre(MeanStdPost):
    def input_transform(self):
        """per-channel form only when all four of mean / std (in, out) are flat lists of one length (``mean2tensor`` also accepts
        scalars-as-tensors and full tensors, which broadcast differently: those run on the host as the reference does)"""
        flat = [x for pair in self.raw for x in pair]
        if not all(isinstance(x, (list, tuple)) and x and all(isinstance(v, (int, float)) for v in x) for x in flat):
            return None
        if len({len(x) for x in flat}) != 1:
            return None
        scale, shift = self.affine()
        return ("affine", scale, shift)

    def preprocess(self, tensor, _outputmodel):
        if isinstance(tensor, list):
            return [self.preprocess(x, _outputmodel) for x in tensor]
        return self._adapt(tensor), None

    def postprocess(self, tensor, outputmodel, meta):
        return tensor


class CirMultiscaleAggregation(Wrapper):
    """Image pyramid in, generalized-mean aggregation of the per-scale descriptors out."""

    PRESETS = {"True": True, "False": False, "ms": True, "ss": False,
               "sms5": [1, 1. / np.sqrt(2), np.sqrt(2), 1. / 2, 2], "sms": [1, 1. / np.sqrt(2), np.sqrt(2)]}

    def __init__(self, scales, device):
        super().__init__(device)
        if isinstance(scales, str):
            scales = self.PRESETS[scales]
        if isinstance(scales, bool):
            scales = [1, 1. / np.sqrt(2), 1. / 2] if scales else [1]
        self.scales = scales
        self.device = torch.device(device) if device is not None else torch.device("cpu")

    def _pyramid(self, single):
        meta = None
        if hasattr(single, "metadata") and hasattr(single, "tensor"):
            single, meta = single.tensor, single.metadata
        if self.device.type == "cuda":
            levels = [ScaledInput(single, float(s)) for s in self.scales]       # resized inside the HIP pack kernel
        else:
            levels = [F.interpolate(single, scale_factor=s, mode="bilinear", align_corners=False) for s in self.scales]
        if meta is None:
            return levels
        return [tensors.as_metadata_tensor(l, meta) if isinstance(l, torch.Tensor) else l for l in levels]

    def preprocess(self, tensor, _outputmodel):
        if len(self.scales) == 1:
            return (tensor if isinstance(tensor, list) else [tensor]), isinstance(tensor, list)
        if isinstance(tensor, list):
            return [lvl for single in tensor for lvl in self._pyramid(single)], True
        return self._pyramid(tensor), False

    @staticmethod
    def aggregate_tensor(tensor, nscales, outputdim, msp):
        assert len(tensor) == nscales, "%s != %s" % (len(tensor), nscales)
        cols = [t.reshape(outputdim, -1) for t in tensor]            # each D x N
        if _on_hip(cols[0]):
            from ... import engine
            v = engine.ms_aggregate(torch.stack([c.t() for c in cols]), msp).t()      # D x N
        else:
            v = torch.zeros_like(cols[0])
            for c in cols:
                v += c.pow(msp)
            v = (v / nscales).pow(1. / msp)
            v = v / v.norm(dim=0, keepdim=True)          # no eps, as the reference
        return v.squeeze(1) if v.shape[1] == 1 else v

    def postprocess(self, tensor, outputmodel, waslist):
        msp = 1
        if len(self.scales) > 1 and outputmodel.meta.get("pooling", None) == "gem" \
                and not outputmodel.meta["regional"] and not outputmodel.meta["whitening"]:
            msp = _scalar_of(outputmodel.pool.p)
        ns, dim = len(self.scales), outputmodel.meta["out_channels"]
        if not waslist:
            return self.aggregate_tensor(tensor, ns, dim, msp)
        assert len(tensor) % ns == 0, "%s %% %s != 0" % (len(tensor), ns)
        return [self.aggregate_tensor(tensor[i:i + ns], ns, dim, msp) for i in range(0, len(tensor), ns)]

    def __repr__(self):
        return "%s(scales=%s)" % (type(self).__name__, self.scales)


_SCALARS = {}


def _scalar_of(param):
    """``param.item()`` (the reference reads the GeM exponent this way on every call, wrapper.py:248-251) without a device -> host synchronisation per call:
    the value is read once per (storage, version) of the parameter.  On a HIP device ``.item()`` waits for every forward queued before it, so the host could
    not issue the aggregation / whitening launches -- nor the next call's -- while the device was still busy (config 4: 0.85 ms of 8 ms per call)."""
    if not param.is_cuda:
        return param.item()
    key = (param.data_ptr(), param._version, str(param.device))
    v = _SCALARS.get(key)
    if v is None:
        if len(_SCALARS) > 64:
            _SCALARS.clear()
        v = _SCALARS[key] = param.item()
    return v


class FakeBatch(Wrapper):
    """list of D x 1 descriptors -> D x len tensor"""

    def postprocess(self, tensor, outputmodel, _meta):
        if not isinstance(tensor, list) or not isinstance(tensor[0], torch.Tensor):
            return tensor
        out = torch.zeros(outputmodel.meta["out_channels"], len(tensor), device=tensor[0].device)
        for j, vec in enumerate(tensor):
            out[:, j] = vec.squeeze()
        return out

    def __repr__(self):
        return "%s()" % type(self).__name__


class CirFakeTupleBatch(FakeBatch):
    @classmethod
    def unsqueeze(cls, tensor):
        if isinstance(tensor, list):
            return [cls.unsqueeze(x) for x in tensor]
        if len(tensor.shape) == 3:
            return tensor.unsqueeze_(0)
        if len(tensor.shape) == 4:
            return tensor
        raise ValueError("Unsupported tensor dimensionality %s" % len(tensor.shape))

    def preprocess(self, tensor, _outputmodel):
        if not isinstance(tensor, list) or not isinstance(tensor[0], list):
            return tensor, False
        width = len(tensor[0])
        flat = []
        for tpl in tensor:
            assert width == len(tpl)
            flat += tpl
        return flat, width


class CirtorchWhiten(Wrapper):
    """Learned whitening {'P': DxD, 'm': Dx1} with optional dimensionality reduction."""

    def __init__(self, whitening, dimensions, device):
        super().__init__(device)
        if isinstance(whitening, str):
            whitening = utils.fs_load_pickle(whitening)
        self.P = torch.tensor(whitening["P"], dtype=torch.float32, device=device)
        self.m = torch.tensor(whitening["m"], dtype=torch.float32, device=device)
        self.dimensions = dimensions or self.P.shape[0]

    def postprocess(self, tensor, _outputmodel, _meta):
        if isinstance(tensor, list):
            return [self.postprocess(t, _outputmodel, _meta) for t in tensor]
        v = tensor.reshape(self.P.shape[1], -1)                       # D x N
        if _on_hip(v):
            from ... import engine
            X = engine.whiten(v.t(), self.P, self.m, self.dimensions).t()
        else:
            X = self.P[:self.dimensions, :].mm(v - self.m)
            X = X / (torch.norm(X, p=2, dim=0, keepdim=True) + 1e-6)
        return X.squeeze(1) if X.shape[1] == 1 else X

    def __repr__(self):
        return "%s(dimensions=%s)" % (type(self).__name__, self.dimensions)


class ClahePost(Wrapper):
    """CLAHE on the network output (wrapper.py:325-348).  Tensors on a HIP device go through gandtr_amd.clahe (whole batch, three
    launches, no host round trip); host tensors take the reference's cv2 route, which needs opencv (optional)."""

    def __init__(self, meanstd, clip_limit=4, grid_size=8, colorspace="lab", *, device):
        super().__init__(device)
        from .transform import ImageClahe
        self.raw = json.loads(meanstd)                   # host copies for the HIP route (no device -> host sync per call)
        self.meanstd = [MeanStdPost.mean2tensor(x, device) for x in self.raw]
        self.clahe = ImageClahe(clip_limit=float(clip_limit), grid_size=int(grid_size), colorspace=colorspace)

    def postprocess(self, tensor, outputmodel, meta):
        if tensor is None:
            return tensor
        if isinstance(tensor, list):
            return [self.postprocess(x, outputmodel, meta) for x in tensor]
        if _on_hip(tensor) and tensor.dim() in (3, 4):
            if self.clahe.colorspace.lower() != "lab":
                raise NotImplementedError("Colorspace %s is not supported on the HIP path" % self.clahe.colorspace)
            from ... import clahe
            pair = (self.raw[0], self.raw[1])
            batch = tensor if tensor.dim() == 4 else tensor[None]
            out = clahe.clahe_lab(batch, self.clahe.clip_limit, self.clahe.grid_size, pair, pair)
            return out if tensor.dim() == 4 else out[0]
        if tensor.dim() == 4:
            return torch.stack([self.postprocess(x, outputmodel, meta) for x in tensor])
        if tensor.dim() == 3:
            t = tensor.detach().mul(self.meanstd[1]).add(self.meanstd[0])
            img = self.clahe.apply(t.cpu().numpy().transpose((1, 2, 0)))
            t = torch.from_numpy(img.transpose((2, 0, 1))).to(tensor.device)
            return t.sub(self.meanstd[0]).div(self.meanstd[1])
        raise ValueError("Unsupported tensor dims: %s" % tensor.dim())


class RgbToBgrPre(Wrapper):
    def input_transform(self):
        return ("perm", [2, 1, 0])

    def preprocess(self, tensor, _outputmodel):
        if isinstance(tensor, list):
            return [self.preprocess(x, _outputmodel) for x in tensor], None
        if tensor.dim() == 4:
            return tensor[:, [2, 1, 0], ...], None
        if tensor.dim() == 3:
            return tensor[[2, 1, 0], ...], None
        raise ValueError("Unsupported tensor dims: %s" % (tensor.shape,))


WRAPPERS_LABELS = {
    "meanstd_post": MeanStdPost,
    "meanstd_pre": MeanStdPre,
    "cirmultiscale": CirMultiscaleAggregation,
    "fakebatch": FakeBatch,
    "cirfaketuplebatch": CirFakeTupleBatch,
    "cirwhiten": CirtorchWhiten,
    "clahepost": ClahePost,
    "rgb2bgr_pre": RgbToBgrPre,
}


def initialize_wrappers(net_wrappers, device):
    """None -> no wrappers; 'name:arg:arg, name2' mini-DSL; or {'<order>_<name>': {kwargs}} sorted by key."""
    if net_wrappers is None:
        wraps = []
    elif isinstance(net_wrappers, str):
        wraps = []
        for wrap in [x.strip() for x in utils.splitp(net_wrappers, ",", check_valid_pairs=True) if x]:
            if not wrap:
                continue
            wname, *args = utils.splitp(wrap, ":")
            wraps.append(WRAPPERS_LABELS[wname](*args, device=device))
    else:
        wraps = [WRAPPERS_LABELS[k.split("_", 1)[1]](**net_wrappers[k], device=device) for k in sorted(net_wrappers)]
    return Compose(wraps, device)
