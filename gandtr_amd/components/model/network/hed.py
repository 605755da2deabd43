"""HED edge detector with bilinear score up-sampling -- host mirror of mdir/components/model/network/hed.py:19-83.
Used forward-only after the hedngan generator (BASELINE config 3; reference call site
mdir/learning/epoch_iteration/edges_epochs.py:87)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ._hipbacked import HipBacked

BLOCKS = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))


class HedInterpolation(HipBacked, nn.Module):
    meta = {"in_channels": 3, "out_channels": 1}
    #: on a HIP device the wrapper chain hands its trailing per-channel input wrappers over as ``input_transform``
    accepts_input_transform = True

    def __init__(self, pretrained=None):
        super().__init__()
        cin = 3
        for i, chans in enumerate(BLOCKS):
            layers = [] if i == 0 else [nn.MaxPool2d(kernel_size=2, stride=2)]
            for c in chans:
                layers += [nn.Conv2d(cin, c, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = c
            setattr(self, "vgg%d" % (i + 1), nn.Sequential(*layers))
        for i, chans in enumerate(BLOCKS):
            setattr(self, "score%d" % (i + 1), nn.Conv2d(chans[-1], 1, kernel_size=1))
        self.fusion = nn.Sequential(nn.Conv2d(5, 1, kernel_size=1))
        if pretrained:
            from ....tools.utils import fs_open
            with fs_open(pretrained) as handle:
                self.load_state_dict(torch.load(handle, map_location="cpu"))

    def forward(self, x, no_sigmoid=False, input_transform=None):
        """``input_transform``: per-channel (perm, scale, shift) applied inside the HIP input-pack kernel, passed per call by Compose
        when the trailing wrappers are RgbToBgrPre / MeanStdPre (components/data/wrapper.py, _fold_input_wrappers)"""
        if input_transform is not None and self._hip_device().type != "cuda":
            raise ValueError("input_transform is a HIP-path argument")
        if self._hip_device().type == "cuda":
            from .... import engine
            self._hip_check_inference()
            prec = self._hip_precision()
            tr = input_transform
            net = self._hip_net(("hed", bool(no_sigmoid), prec, tr),
                                lambda sd, dev: engine.build_hed(sd, dev, sigmoid=not no_sigmoid, precision=prec, perm=None if tr is None else list(tr[0]),
                                                                 in_affine=None if tr is None else (list(tr[1]), list(tr[2]))))
            return net.forward(x)[net.out_slot]
        size = (x.size(2), x.size(3))
        feats, h = [], x
        for i in range(5):
            h = getattr(self, "vgg%d" % (i + 1))(h)
            feats.append(h)
        scores = [F.interpolate(getattr(self, "score%d" % (i + 1))(f), size=size, mode="bilinear", align_corners=False)
                  for i, f in enumerate(feats)]
        o = self.fusion(torch.cat(scores, 1))
        return o if no_sigmoid else torch.sigmoid(o)
