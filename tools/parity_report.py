"""dev (GPU box): generator parity numbers in the default (f16c) precision -- pre-tanh relative error and IMAGE-level absolute error (max,
p99.9, mean) per weight set and batch, plus the hub's seed-0 fixtures.  The gates of tests/test_hip_models.py / test_hip_golden.py and
the `parity` string of bench.py quote these.  usage: python tools/parity_report.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import hubconf
from gandtr_amd.engine import build_generator
from gandtr_amd.tools import synth
from oracle import gandtr_oracle as O

dev = torch.device("cuda:0")
rows = []


def stats(d):
    d = d.abs().flatten()
    return {"max": float(d.max()), "p999": float(torch.quantile(d[:: max(1, d.numel() // 4000000)], 0.999)), "mean": float(d.mean())}


def head_scale(sd, factor):
    sd = dict(sd)
    sd["model.26.weight"] = sd["model.26.weight"] * factor
    sd["model.26.bias"] = sd["model.26.bias"] * factor
    return sd


for name, norm, gain, scale in [("instance gain 0.02", "instance", 0.02, None), ("instance gain 0.2", "instance", 0.2, None),
                                ("batch (kaiming)", "batch", None, None), ("batch, head scaled to max|pre-tanh| = 3", "batch", None, 3.0),
                                ("instance gain 0.2, head scaled to max|pre-tanh| = 3", "instance", 0.2, 3.0)]:
    for batch in (2, 8):
        sd = synth.generator_state(0, norm, gain=gain or 0.02)
        x = synth.synth_input(2, (batch, 3, 256, 256), 1.0)
        if scale:
            _, f = O.resnet_generator(x, sd, norm, 9, taps=(26,))
            sd = head_scale(sd, scale / float(f[26].abs().max()))
        ref, feats = O.resnet_generator(x, sd, norm, 9, taps=(26,))
        net = build_generator(sd, dev, taps=(26,))
        outs = net.forward(x.to(dev))
        pre, img = outs[net.tap_slots[26]].cpu(), outs[net.out_slot].cpu()
        r = {"weights": name, "batch": batch, "max_abs_pre_tanh": float(feats[26].abs().max()),
             "pre_tanh_rel": float((pre - feats[26]).abs().max() / feats[26].abs().max()), "image_abs": stats(img - ref)}
        rows.append(r)
        print(json.dumps(r))
for hub in ("cyclegan", "hedngan"):
    g = np.load(os.path.join("tests", "golden", "hub_%s.npz" % hub))
    net = getattr(hubconf, hub)(pretrained=False, device=dev)
    x = synth.synth_input(3, (4, 3, 256, 256), 1.0)
    with torch.no_grad():
        y = net(x).cpu()
    r = {"weights": "hub %s (seed-0 init, reference output fixture, 1/64 sub-sample)" % hub, "batch": 4,
         "image_abs": stats(y[:, :, ::8, ::8] - torch.from_numpy(g["out_sub"]))}
    rows.append(r)
    print(json.dumps(r))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(rows, f, indent=1)
