"""torch.hub manifest -- same entrypoints as the reference's hubconf.py:1-4."""
from gandtr_amd.hub.model import gem_vgg16_cyclegan, gem_vgg16_hedngan, gem_resnet101_cyclegan, gem_resnet101_hedngan, \
    hedngan, cyclegan

dependencies = ["torch"]
