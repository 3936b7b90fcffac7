"""ResNet-18 feature extractor with timm's module naming (third-party timm==0.9.7 `resnet18`,
features_only=True; call sites muvo/models/mile.py:24-26,81-83 and common.py:15), on the HIP conv/BN kernels.
conv -> train-mode BN -> ReLU are issued as conv kernel + fused BN(+residual)+ReLU kernel."""
import torch.nn as nn

from muvo_amd import nn as hnn
from muvo_amd import ops
from muvo_amd.layers.layers import BasicBlock

_FEATURE_INFO = [dict(num_chs=64, reduction=2), dict(num_chs=64, reduction=4), dict(num_chs=128, reduction=8),
                 dict(num_chs=256, reduction=16), dict(num_chs=512, reduction=32)]


class FeatureInfo:
    def __init__(self, infos):
        self.infos = infos

    def get_dicts(self, keys=None):
        return [{k: d[k] for k in (keys or d.keys())} for d in self.infos]


class ResNet18Features(nn.Module):
    def __init__(self, in_chans=3, out_indices=(2, 3, 4)):
        super().__init__()
        self.conv1 = hnn.Conv2d(in_chans, 64, 7, 2, 3, bias=False)
        self.bn1 = hnn.BatchNorm2d(64)
        inplanes = 64
        for i, planes in enumerate((64, 128, 256, 512)):
            stride = 1 if i == 0 else 2
            setattr(self, f'layer{i + 1}', nn.Sequential(
                BasicBlock(inplanes, planes, stride=stride, downsample=True if stride != 1 else None),
                BasicBlock(planes, planes)))
            inplanes = planes
        self.out_indices = tuple(out_indices)
        self.feature_info = FeatureInfo([_FEATURE_INFO[i] for i in self.out_indices])

    def forward(self, x, feat_consumers=None):
        """feat_consumers: per returned feature map (order of out_indices) the convolution modules outside this network that read
        it (the skip convolutions of DecoderDS): the last BatchNorm of that stage writes their split planes too."""
        feats = []
        x = self.bn1(self.conv1(x), relu=True, from_conv=True)
        feats.append(x)
        x = ops.max_pool2d(x, 3, 2, 1)
        blocks = [b for i in range(4) for b in getattr(self, f'layer{i + 1}')]
        ext = {idx: tuple(feat_consumers[k]) for k, idx in enumerate(self.out_indices)} if feat_consumers is not None else {}
        for j, blk in enumerate(blocks):
            nxt = (blocks[j + 1].conv1,) if j + 1 < len(blocks) else ()
            if j % 2 == 1:                      # last block of stage j // 2 + 1 = feature index j // 2 + 1
                nxt = nxt + ext.get(j // 2 + 1, ())
            x = blk(x, next_convs=nxt)
            if j % 2 == 1:
                feats.append(x)
        return [feats[i] for i in self.out_indices]
