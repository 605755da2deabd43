"""``infer`` stage -- signature and loop shape of mdir/stages/infer.py:17-66.

The reference body wires datasets (DataLoader with 6 workers), output sinks and resource statistics around
``out = network(indata)``; those subsystems are out of scope, so this mirror accepts in-memory
inputs: ``data[0]`` is a sequence of image tensors (C x H x W or 1 x C x H x W).  Per item it calls ``network(indata)`` under
``torch.no_grad()`` exactly like the reference (batch size 1, the only mode the reference's wrappers support) and collects the
outputs according to ``params["output"]["inference"]["name"]``:
    "embedding" -> one (N x D) float32 numpy array   (EmbeddingOutput, mdir/components/data/output.py:118-156)
    "rgb"       -> a list of H x W x 3 float arrays in [0, 1] (un-normalised with the network's mean_std; RgbImageSaver :75-84)
"""
import copy
import time

import numpy as np
import torch

from ..learning import load_network


def infer(params, data):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    np.random.seed(0)
    torch.manual_seed(0)
    if not len(data[0]):
        return ({"status": "skipped"},)
    network = load_network(copy.deepcopy(params["network"]), device).eval()
    kind = params["output"]["inference"]["name"]
    if kind not in ("embedding", "rgb"):
        raise KeyError(kind)
    mean_std = network.network_params.runtime.get("data", {}).get("mean_std")
    outputs, t0 = [], time.time()
    with torch.no_grad():
        forward = getattr(network, params["forward"]["method"]) if "forward" in params else network
        for indata in data[0]:
            indata = torch.as_tensor(indata)
            if indata.dim() == 3:
                indata = indata.unsqueeze(0)
            if "forward" in params:
                out = forward(indata.to(device), **params["forward"]["params"])
            else:
                out = network(indata)
            if kind == "embedding":
                outputs.append(out.detach().float().cpu().numpy().reshape(-1))
            else:
                img = out.detach().float().cpu()[0]
                if mean_std is not None:
                    img = img * torch.tensor(mean_std[1])[:, None, None] + torch.tensor(mean_std[0])[:, None, None]
                outputs.append(img.clamp(0, 1).permute(1, 2, 0).numpy())
    metadata = {"stats": {"items": len(outputs), "seconds": time.time() - t0}}
    if kind == "embedding":
        return (metadata, np.stack(outputs))
    return (metadata, outputs)
