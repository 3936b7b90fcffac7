"""Scaffolding stub: standard ResNet-18 with timm's module names, features_only interface
(timm==0.9.7 semantics: FeatureListNet keeps conv1/bn1/act1/maxpool/layer1-4 flat)."""
import torch.nn as nn


def downsample_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None,
                    norm_layer=None):
    norm_layer = norm_layer or nn.BatchNorm2d
    kernel_size = 1 if stride == 1 and dilation == 1 else kernel_size
    first_dilation = (first_dilation or dilation) if kernel_size > 1 else 1
    p = ((stride - 1) + first_dilation * (kernel_size - 1)) // 2
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=p, dilation=first_dilation,
                  bias=False),
        norm_layer(out_channels))


class _Block(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        shortcut = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        if self.downsample is not None:
            shortcut = self.downsample(shortcut)
        x = x + shortcut
        return self.act2(x)


class _FeatureInfo:
    def __init__(self, infos):
        self._infos = infos

    def get_dicts(self, keys=None):
        return [{k: d[k] for k in keys} for d in self._infos]


class _ResNet18Features(nn.Module):
    def __init__(self, in_chans=3, out_indices=(2, 3, 4)):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        chans = [64, 128, 256, 512]
        inpl = 64
        for i, c in enumerate(chans):
            stride = 1 if i == 0 else 2
            ds = downsample_conv(inpl, c, 1, stride) if (stride != 1 or inpl != c) else None
            setattr(self, f'layer{i + 1}', nn.Sequential(_Block(inpl, c, stride, ds), _Block(c, c)))
            inpl = c
        self.out_indices = tuple(out_indices)
        allinfo = [dict(num_chs=64, reduction=2), dict(num_chs=64, reduction=4), dict(num_chs=128, reduction=8),
                   dict(num_chs=256, reduction=16), dict(num_chs=512, reduction=32)]
        self.feature_info = _FeatureInfo([allinfo[i] for i in self.out_indices])

    def forward(self, x):
        feats = []
        x = self.act1(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.maxpool(x)
        for i in range(4):
            x = getattr(self, f'layer{i + 1}')(x)
            feats.append(x)
        return [feats[i] for i in self.out_indices]


def create_model(name, pretrained=False, features_only=False, out_indices=(2, 3, 4), in_chans=3, **kw):
    assert name == 'resnet18' and features_only
    return _ResNet18Features(in_chans=in_chans, out_indices=out_indices)
