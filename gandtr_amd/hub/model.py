"""torch.hub entrypoints -- drop-in for mdir/hub/model.py:17-154 (re-exported by /hubconf.py).

The two YAML scenarios the reference reads (mdir/hub/generator.yml, mdir/hub/embedding.yml) are small static
configurations; they are kept here as Python dicts (fresh deep copy per call -- the reference's ``_create`` mutates
its parsed YAML in place).
"""
import copy

import torch

from ..components.data.transform import initialize_transforms
from ..learning.checkpoints import Checkpoints
from ..learning.network import initialize_network

BASE_URL = "http://ptak.felk.cvut.cz/personal/jenicto2/download/iccv23_gan/"

GENERATOR_SCENARIO = {
    "initialized": {
        "type": "SingleNetwork",
        "model": {"architecture": "official_resnet_generator", "no_antialias": True, "no_antialias_up": True,
                  "input_nc": 3, "output_nc": 3, "n_blocks": 9, "norm_layer": "instance"},
        "initialize": {"weights": "normal_p2p", "seed": 0},
        "runtime": {"wrappers": "",
                    "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5, 0.5, 0.5], [0.5, 0.5, 0.5]]}},
    },
    "pretrained": {"path": None, "runtime": {"wrappers": ""}},
}

EMBEDDING_SCENARIO = {
    "initialized": {
        "type": "SingleNetwork",
        "model": {"architecture": "cirnet", "cir_architecture": None, "local_whitening": False, "pooling": "gem",
                  "pretrained": False, "regional": False, "whitening": False},
        "initialize": False,
        "runtime": {"data": {"transforms": "pil2np | apply_clahe:1.0 | totensor | normalize",
                             "mean_std": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]},
                    "wrappers": "cirfaketuplebatch"},
    },
    "pretrained": {
        "path": None,
        "runtime": {"wrappers": {"train": None,
                                 "eval": {"0_cirwhiten": {"whitening": None, "dimensions": None},
                                          "1_cirmultiscale": {"scales": True}}}},
    },
}


def _create(scenario, substitutions, pretrained, device):
    params = copy.deepcopy(scenario["pretrained" if pretrained else "initialized"])
    for target, value in substitutions.items():
        p = params
        *parents, leaf = target.split(".")
        for k in parents:
            p = p[k]
        p[leaf] = value
    if not device:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if pretrained:
        state = Checkpoints.load_network(params["path"])
        if state["net"]["network_params"]["model"]["architecture"] == "cirnet":
            state["net"]["network_params"]["model"]["pretrained"] = False
        network = initialize_network(None, device, state, params["runtime"]).eval()
    else:
        network = initialize_network(params, device).eval()
    data_params = network.network_params.runtime["data"]
    if "augmentations" not in data_params:
        data_params["augmentations"] = data_params.pop("transforms")
    network.transform = initialize_transforms(**data_params)
    # device-side counterpart for decoded uint8 images already in HBM (no reference equivalent: SURVEY.md section 8f, ingest row)
    from ..ingest import DeviceTransform
    network.transform_device = DeviceTransform(**data_params)
    return network


def _embedding(name, arch, pretrained, device):
    if pretrained:
        return _create(EMBEDDING_SCENARIO, {"path": "%s%s.pth" % (BASE_URL, name),
                                            "runtime.wrappers.eval.0_cirwhiten.whitening": "%s%s_lw.pkl" % (BASE_URL, name)},
                       pretrained, device)
    return _create(EMBEDDING_SCENARIO, {"model.cir_architecture": arch}, pretrained, device)


def gem_vgg16_cyclegan(pretrained=True, device=None):
    """GeM descriptor, VGG16 trunk, trained with CycleGAN day->night query augmentation (expects CLAHE'd input)."""
    return _embedding("cyclegan_embed_vgg16", "vgg16", pretrained, device)


def gem_vgg16_hedngan(pretrained=True, device=None):
    """GeM descriptor, VGG16 trunk, trained with HED-N-GAN query augmentation (expects CLAHE'd input)."""
    return _embedding("hedngan_embed_vgg16", "vgg16", pretrained, device)


def gem_resnet101_cyclegan(pretrained=True, device=None):
    """GeM descriptor, ResNet-101 trunk, trained with CycleGAN query augmentation (expects CLAHE'd input)."""
    return _embedding("cyclegan_embed_resnet101", "resnet101", pretrained, device)


def gem_resnet101_hedngan(pretrained=True, device=None):
    """GeM descriptor, ResNet-101 trunk, trained with HED-N-GAN query augmentation (expects CLAHE'd input)."""
    return _embedding("hedngan_embed_resnet101", "resnet101", pretrained, device)


def cyclegan(pretrained=True, device=None):
    """ResNet CycleGAN day->night generator (InstanceNorm)."""
    if pretrained:
        return _create(GENERATOR_SCENARIO, {"path": BASE_URL + "cyclegan_generator_X.pth"}, pretrained, device)
    return _create(GENERATOR_SCENARIO, {}, pretrained, device)


def hedngan(pretrained=True, device=None):
    """ResNet HED-N-GAN day->night generator (BatchNorm, kaiming init when not pretrained)."""
    if pretrained:
        return _create(GENERATOR_SCENARIO, {"path": BASE_URL + "hedngan_generator_X.pth"}, pretrained, device)
    return _create(GENERATOR_SCENARIO, {"model.norm_layer": "batch", "initialize.weights": "kaiming_p2p"}, pretrained, device)
