"""Device-side counterpart of the dataset the evaluation feeds the network from: ``ImagesFromList``
(mdir/external/cirtorch/datasets/genericdataset.py:12-105).

Same constructor arguments and the same per-image steps -- load (``pil_loader``), optional crop to the query bounding box, ``imresize`` to
``imsize`` (scaled by the crop's share of the full image, :86-91), transform -- but every step runs on the device: JPEG decoding
(gandtr_amd/jpeg.py), the crop is a view of the decoded tensor, resize / [0, 1] scaling / CLAHE / normalisation are the ingest launches
(gandtr_amd/ingest.py).  ``transform`` is a ``DeviceTransform`` (what the hub attaches to a network as ``.transform_device``) or a
``(mean, std)`` pair.  ``loader`` (optional) is the caller's own ``bytes -> H x W x 3 uint8 array`` function for files the device decoder
refuses (progressive JPEG, PNG ...): without it such a file raises.  ``batch(indices)`` decodes and resizes a whole list of items in one
set of launches (the reference's DataLoader walks the items one by one in worker processes); ``__getitem__`` is ``batch([i])[0]``.
Not mirrored: HDF5 roots and ``load_images_with_bbx`` (pre-cropped files on disk)."""
import os

import torch

from . import ingest, jpeg


class ImagesFromList:
    def __init__(self, root, images, imsize=None, bbxs=None, transform=None, loader=None, ignore_errors=False, load_images_with_bbx=False,
                 image_labels=None, device=None):
        if load_images_with_bbx:
            raise NotImplementedError("load_images_with_bbx (pre-cropped files) is not mirrored")
        if root and str(root).endswith(".h5"):
            raise NotImplementedError("HDF5 image stores are not mirrored")
        images = list(images)
        if len(images) == 0:
            raise RuntimeError("Dataset contains 0 images!")                    # genericdataset.py:54-55
        self.root, self.images, self.imsize, self.bbxs = root, images, imsize, bbxs
        self.images_fn = [img if isinstance(img, (bytes, bytearray)) else (os.path.join(root, img) if root else img) for img in images]
        self.transform, self.loader, self.ignore_errors, self.image_labels = transform, loader, ignore_errors, image_labels
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def __len__(self):
        return len(self.images_fn)

    def _transform_args(self):
        t = self.transform
        if t is None:
            return [0.0] * 3, [1.0] * 3, None, 8
        if isinstance(t, ingest.DeviceTransform):
            mean, std = (t.mean, t.std) if t.normalize else ([0.0] * 3, [1.0] * 3)
            return mean, std, t.clahe_clip, t.clahe_grid
        mean, std = t
        return [float(v) for v in mean], [float(v) for v in std], None, 8

    def batch(self, indices):
        """items ``indices`` as fp32 3 x h x w device tensors (uint8 H x W x 3 when neither ``imsize`` nor ``transform`` is set)"""
        indices = list(indices)
        try:
            decoded = jpeg.load_many([self.images_fn[i] for i in indices], self.device, host_loader=self.loader)
        except (OSError, ValueError):
            if not self.ignore_errors:
                raise
            return [self._one_or_empty(i) for i in indices]
        sizes = []
        for k, i in enumerate(indices):
            img = decoded[k]
            full = max(img.shape[0], img.shape[1])                              # imfullsize = max(img.size)
            box = self.bbxs[i] if self.bbxs is not None and self.bbxs[i] else None
            if box:
                x0, y0, x1, y1 = (int(round(v)) for v in box)                   # Image.crop rounds the box; parts outside the image are not supported here
                if not (0 <= x0 < x1 <= img.shape[1] and 0 <= y0 < y1 <= img.shape[0]):
                    raise ValueError("bounding box %s outside the %d x %d image" % (tuple(box), img.shape[1], img.shape[0]))
                img = img[y0:y1, x0:x1].contiguous()
                decoded[k] = img
            if self.imsize is None:
                sizes.append(max(img.shape[0], img.shape[1]))                   # thumbnail to its own size: unchanged
            else:
                sizes.append(self.imsize * max(img.shape[0], img.shape[1]) / full if box else self.imsize)
        if self.imsize is None and self.transform is None:
            return decoded
        mean, std, clip, grid = self._transform_args()
        return ingest.ingest_many(decoded, sizes, mean, std, clip, grid)

    def _one_or_empty(self, i):
        try:
            return self.batch_strict([i])[0]
        except (OSError, ValueError):
            return {}                                                            # genericdataset.py:71-73

    def batch_strict(self, indices):
        keep, self.ignore_errors = self.ignore_errors, False
        try:
            return self.batch(indices)
        finally:
            self.ignore_errors = keep

    def __getitem__(self, index):
        return self.batch([index])[0]

    def __repr__(self):
        return "Dataset %s\n    Number of images: %d\n    Root Location: %s\n    Transforms (if any): %r\n" % (
            type(self).__name__, len(self), self.root, self.transform)
