"""Scaffolding stub: the torchvision.transforms surface the reference touches at import/ctor time."""
import torch.nn as nn
from . import functional  # noqa
from .functional import InterpolationMode  # noqa


class Normalize(nn.Module):
    def __init__(self, mean=None, std=None, inplace=False):
        super().__init__()
        self.mean, self.std = mean, std


class _Identity(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        return x


RandomApply = _Identity
ColorJitter = _Identity
RandomAffine = _Identity
