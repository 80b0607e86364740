"""ORACLE (test infrastructure, never on the product path).

DenseNet-121 restated from the published architecture, because the reference
takes it from an un-vendored third-party dependency:

    torchvision.models.densenet.densenet121(pretrained=True)
    constructed at /root/reference/code/models.py:22-24 and :308-310,
    used only as `<trunk>.features(x)` (models.py:384-385 and 17 sibling sites).

torchvision is NOT installed in this image and no version is pinned by the
reference (no requirements.txt; README.md:22 says "PyTorch 1.0+").  The
architecture below is the published densenet121 configuration: growth_rate 32,
block_config (6, 12, 24, 16), num_init_features 64, bn_size 4, drop_rate 0,
with torchvision's state-dict key names so reference
snapshots load:

    features.conv0 / norm0 / (relu0) / (pool0)
    features.denseblock{K}.denselayer{L}.{norm1,conv1,norm2,conv2}
    features.transition{K}.{norm,conv}       (+ relu, pool without parameters)
    features.norm5
    classifier  (Linear 1024 -> 1000; never executed by the reference)

Parity status: "parity unpinned" by the reference itself (it has no tests and
ImageNet weights are unobtainable offline); pinned instead against outputs of the
reference's own Python run in the build container through oracle/make_golden.py.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

GROWTH = 32
BLOCK_CONFIG = (6, 12, 24, 16)
INIT_FEATURES = 64
BN_SIZE = 4


class _DenseLayer(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, BN_SIZE * GROWTH, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(BN_SIZE * GROWTH)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(BN_SIZE * GROWTH, GROWTH, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1)
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, n_layers, cin):
        super().__init__()
        for i in range(n_layers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(cin + i * GROWTH)

    def forward(self, x):
        feats = [x]
        for _, layer in self.items():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv = nn.Conv2d(cin, cout, kernel_size=1, stride=1, bias=False)
        self.pool = nn.AvgPool2d(kernel_size=2, stride=2)


class DenseNet121(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv2d(3, INIT_FEATURES, kernel_size=7, stride=2, padding=3, bias=False)),
            ("norm0", nn.BatchNorm2d(INIT_FEATURES)),
            ("relu0", nn.ReLU(inplace=True)),
            ("pool0", nn.MaxPool2d(kernel_size=3, stride=2, padding=1)),
        ]))
        c = INIT_FEATURES
        for k, n_layers in enumerate(BLOCK_CONFIG):
            self.features.add_module("denseblock%d" % (k + 1), _DenseBlock(n_layers, c))
            c += n_layers * GROWTH
            if k != len(BLOCK_CONFIG) - 1:
                self.features.add_module("transition%d" % (k + 1), _Transition(c, c // 2))
                c //= 2
        self.features.add_module("norm5", nn.BatchNorm2d(c))
        self.classifier = nn.Linear(c, 1000)

    def forward(self, x):  # never called by the reference; kept for completeness
        f = F.relu(self.features(x), inplace=True)
        f = torch.flatten(F.adaptive_avg_pool2d(f, (1, 1)), 1)
        return self.classifier(f)


def densenet121(pretrained=False, **_):
    """Drop-in for the torchvision constructor the reference calls.

    `pretrained=True` cannot be honoured offline; weights are whatever the caller
    loads afterwards (the golden-vector script loads seeded synthetic weights
    through the reference's own load_state_dict)."""
    return DenseNet121()
