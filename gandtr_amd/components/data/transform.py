"""CPU image preprocessing attached to hub models as ``.transform`` -- host mirror of
mdir/components/data/transform/__init__.py:37-46 (initialize_transforms), core_transforms.py:25-100
(Compose, ToTensor, Normalize, Pil2Numpy) and photometric_transforms.py:28-36 / functional.py:140-161 (ApplyClahe).
Host-side only (out of the HIP scope); CLAHE needs opencv and raises ImportError without it."""
import numpy as np
import torch


class GenericTransform:
    def __init__(self, params=None):
        self.params = params or {}

    def __repr__(self):
        return type(self).__name__ + "(%s)" % ", ".join("%s=%s" % kv for kv in self.params.items())


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, *pics):
        for t in self.transforms:
            pics = t(*pics)
        return pics[0] if len(pics) == 1 else pics

    def __repr__(self):
        return type(self).__name__ + "(" + "".join("\n    %s" % t for t in self.transforms) + "\n)"


class Pil2Numpy(GenericTransform):
    """PIL image / ndarray -> float32 HWC array in [0, 1]"""

    def __call__(self, *pics):
        out = []
        for pic in pics:
            if hasattr(pic, "convert"):
                pic = np.asarray(pic.convert("RGB"))
            elif not isinstance(pic, np.ndarray):
                raise ValueError("Unsupported type '%s'" % type(pic))
            if pic.dtype == np.uint8:
                pic = pic.astype(np.float32) / 255.0
            elif pic.dtype == np.uint16:
                pic = pic.astype(np.float32) / 65535.0
            else:
                pic = pic.astype(np.float32)
            out.append(pic)
        return out


class ImageClahe:
    """CLAHE on the L channel of LAB (8-bit quantised), cv2-based as in the reference (functional.py:140-161)."""

    def __init__(self, clip_limit, grid_size, colorspace="lab"):
        self.clip_limit, self.grid_size, self.colorspace = clip_limit, grid_size, colorspace

    def apply(self, img):
        if isinstance(img, torch.Tensor) and img.is_cuda:       # H x W x 3 on a HIP device: gandtr_amd/csrc/clahe.hip
            if self.colorspace != "lab":
                raise NotImplementedError("only the 'lab' colorspace is supported")
            from ... import clahe
            return clahe.clahe_lab(img.permute(2, 0, 1)[None], self.clip_limit, self.grid_size)[0].permute(1, 2, 0)
        try:
            import cv2
        except ImportError as e:          # pragma: no cover - depends on the image
            raise ImportError("CLAHE preprocessing needs opencv-python (cv2), which is not installed") from e
        if self.colorspace != "lab":
            raise NotImplementedError("only the 'lab' colorspace is supported")
        lab = cv2.cvtColor(img.astype(np.float32), cv2.COLOR_RGB2LAB)
        clahe = cv2.createCLAHE(clipLimit=self.clip_limit, tileGridSize=(self.grid_size, self.grid_size))
        chan = (lab[..., 0] / 100.0 * 255).astype(np.uint8)
        lab[..., 0] = clahe.apply(chan).astype(np.float32) / 255.0 * 100.0
        return cv2.cvtColor(lab, cv2.COLOR_LAB2RGB)


class ApplyClahe(GenericTransform):
    def __init__(self, clip_limit, grid_size=8, colorspace="lab"):
        super().__init__({"clip_limit": float(clip_limit), "grid_size": int(grid_size), "colorspace": colorspace})
        self.clahe = ImageClahe(**self.params)

    def __call__(self, *pics):
        return [self.clahe.apply(p) for p in pics]


class ToTensor(GenericTransform):
    """HWC float array -> CHW tensor (uint8 input is scaled to [0, 1], like torchvision's ToTensor)"""

    def __call__(self, *pics):
        out = []
        for pic in pics:
            pic = np.asarray(pic)
            if pic.ndim == 2:
                pic = pic[:, :, None]
            t = torch.from_numpy(np.ascontiguousarray(pic.transpose((2, 0, 1))))
            out.append(t.float().div(255) if t.dtype == torch.uint8 else t)
        return out


class Normalize(GenericTransform):
    def __init__(self, mean, std, strict_shape=True):
        if isinstance(strict_shape, str):
            strict_shape = strict_shape.lower() != "false"
        super().__init__({"mean": mean, "std": std, "strict_shape": bool(strict_shape)})
        assert len(mean) == len(std)

    def __call__(self, *pics):
        out = []
        for pic in pics:
            n = pic.size(0)
            if self.params["strict_shape"]:
                assert n == len(self.params["mean"]), (n, len(self.params["mean"]))
            else:
                assert n <= len(self.params["mean"]), (n, len(self.params["mean"]))
            mean = torch.as_tensor(self.params["mean"][:n], dtype=pic.dtype)[:, None, None]
            std = torch.as_tensor(self.params["std"][:n], dtype=pic.dtype)[:, None, None]
            out.append((pic - mean) / std)
        return out


TRANSFORMS = {"totensor": ToTensor, "normalize": Normalize, "pil2np": Pil2Numpy, "apply_clahe": ApplyClahe}


def initialize_transforms(augmentations, mean_std):
    """'pil2np | apply_clahe:1.0 | totensor | normalize' -> Compose"""
    trans = []
    for aug in [x.strip() for x in augmentations.split("|") if x.strip()]:
        tname, *args = aug.split(":", 1)
        args = args[0].split(":") if args else []
        trans.append(TRANSFORMS[tname](*(list(mean_std) + args)) if "normalize" in aug else TRANSFORMS[tname](*args))
    return Compose(trans)
