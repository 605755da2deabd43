"""VGG16 and ResNet-50/101/152 module graphs.

The reference obtains these from ``torchvision.models.<arch>(pretrained=False)`` (mdir/external/cirtorch/networks/
imageretrievalnet.py:174-180) and keeps ``features.children()[:-1]`` (VGG) / ``children()[:-2]`` (ResNet), :185-190.
torchvision is neither vendored by the reference nor installed here, so the published architectures are restated with
torchvision's module names (checkpoint keys ``features.<i>...`` must match): VGG cfg "D"; ResNet v1.5 Bottleneck
(stride on the 3x3 conv, expansion 4).  Parity of this layer graph is pinned by no reference test.
"""
import torch
import torch.nn as nn

VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class VGG(nn.Module):
    def __init__(self, cfg=VGG16_CFG, num_classes=1000):
        super().__init__()
        layers, cin = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096),
                                        nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes))

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        return self.relu(o + idt)


class ResNet(nn.Module):
    def __init__(self, blocks, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, blocks[0], 1)
        self.layer2 = self._make_layer(128, blocks[1], 2)
        self.layer3 = self._make_layer(256, blocks[2], 2)
        self.layer4 = self._make_layer(512, blocks[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def _no_pretrained(pretrained):
    if pretrained:
        raise ValueError("ImageNet-pretrained torchvision weights are not available offline; load a checkpoint instead")


def vgg16(pretrained=False):
    _no_pretrained(pretrained)
    return VGG(VGG16_CFG)


def resnet50(pretrained=False):
    _no_pretrained(pretrained)
    return ResNet((3, 4, 6, 3))


def resnet101(pretrained=False):
    _no_pretrained(pretrained)
    return ResNet((3, 4, 23, 3))


def resnet152(pretrained=False):
    _no_pretrained(pretrained)
    return ResNet((3, 8, 36, 3))


ARCHITECTURES = {"vgg16": vgg16, "resnet50": resnet50, "resnet101": resnet101, "resnet152": resnet152}
