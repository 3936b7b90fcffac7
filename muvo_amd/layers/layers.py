"""BasicBlock (muvo/layers/layers.py:9-66; same block as timm's ResNet BasicBlock) on HIP kernels:
conv3x3 -> BN+ReLU -> conv3x3 -> BN (+shortcut, optionally 1x1-s2 conv + BN) -> ReLU, with the
add + ReLU fused into the second BN kernel."""
import torch.nn as nn

from muvo_amd import nn as hnn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = hnn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = hnn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.downsample = None
        if downsample is not None:
            # timm.models.resnet.downsample_conv(kernel_size=1, stride=2): Conv2d 1x1 s2 p0 (no bias) + BatchNorm2d
            self.downsample = nn.Sequential(hnn.Conv2d(inplanes, planes, 1, 2, 0, bias=False), hnn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x, next_convs=()):
        """next_convs: the convolution modules that will read this block's output (the next block's conv1, a decoder's skip
        convolution): bn2 then writes their split planes along with the fp32 result.  bn1's result has conv2 as its only consumer:
        it exists as planes only when conv2 runs on the bf16x3 kernels; both BatchNorms hand their dx to the convolution in front
        of them as planes (ops.bn_act)."""
        shortcut = x
        y = self.bn1(self.conv1(x), relu=True, consumers=(self.conv2,), sole_consumer=True, from_conv=True)
        y = self.conv2(y)
        if self.downsample is not None:
            shortcut = self.downsample[1](self.downsample[0](shortcut), relu=False, from_conv=True)
        return self.bn2(y, residual=shortcut, res_mode=1, relu=True, consumers=tuple(next_convs), from_conv=True)
