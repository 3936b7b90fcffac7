"""Scaffolding stub (test infrastructure): the torchvision.transforms classes the reference constructs, restated from the
published torchvision 0.15.2 sources (transforms/transforms.py): RandomApply, ColorJitter, RandomAffine — including the order
of their random-number calls, which the product's host-side draw (muvo_amd/augment.py) reproduces."""
import numbers

import torch
import torch.nn as nn

from . import functional  # noqa
from . import functional as F
from .functional import InterpolationMode  # noqa


class Normalize(nn.Module):
    def __init__(self, mean=None, std=None, inplace=False):
        super().__init__()
        self.mean, self.std = mean, std


class RandomApply(nn.Module):
    def __init__(self, transforms, p=0.5):
        super().__init__()
        self.transforms = transforms
        self.p = p

    def forward(self, img):
        if self.p < torch.rand(1):
            return img
        for t in self.transforms:
            img = t(img)
        return img


class ColorJitter(nn.Module):
    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        super().__init__()
        self.brightness = self._check_input(brightness, 'brightness')
        self.contrast = self._check_input(contrast, 'contrast')
        self.saturation = self._check_input(saturation, 'saturation')
        self.hue = self._check_input(hue, 'hue', center=0, bound=(-0.5, 0.5), clip_first_on_zero=False)

    @staticmethod
    def _check_input(value, name, center=1, bound=(0, float('inf')), clip_first_on_zero=True):
        if isinstance(value, numbers.Number):
            if value < 0:
                raise ValueError(f'If {name} is a single number, it must be non negative.')
            value = [center - float(value), center + float(value)]
            if clip_first_on_zero:
                value[0] = max(value[0], 0.0)
        else:
            value = [float(value[0]), float(value[1])]
        if not bound[0] <= value[0] <= value[1] <= bound[1]:
            raise ValueError(f'{name} values should be between {bound}, but got {value}.')
        if value[0] == value[1] == center:
            return None
        return tuple(value)

    @staticmethod
    def get_params(brightness, contrast, saturation, hue):
        fn_idx = torch.randperm(4)
        b = None if brightness is None else float(torch.empty(1).uniform_(brightness[0], brightness[1]))
        c = None if contrast is None else float(torch.empty(1).uniform_(contrast[0], contrast[1]))
        s = None if saturation is None else float(torch.empty(1).uniform_(saturation[0], saturation[1]))
        h = None if hue is None else float(torch.empty(1).uniform_(hue[0], hue[1]))
        return fn_idx, b, c, s, h

    def forward(self, img):
        fn_idx, brightness_factor, contrast_factor, saturation_factor, hue_factor = self.get_params(
            self.brightness, self.contrast, self.saturation, self.hue)
        for fn_id in fn_idx:
            if fn_id == 0 and brightness_factor is not None:
                img = F.adjust_brightness(img, brightness_factor)
            elif fn_id == 1 and contrast_factor is not None:
                img = F.adjust_contrast(img, contrast_factor)
            elif fn_id == 2 and saturation_factor is not None:
                img = F.adjust_saturation(img, saturation_factor)
            elif fn_id == 3 and hue_factor is not None:
                img = F.adjust_hue(img, hue_factor)
        return img


def _setup_angle(x, name, req_sizes=(2,)):
    if isinstance(x, numbers.Number):
        if x < 0:
            raise ValueError(f'If {name} is a single number, it must be positive.')
        x = [-x, x]
    elif len(x) not in req_sizes:
        raise ValueError(f'{name} should be a sequence of length {req_sizes}.')
    return [float(d) for d in x]


class RandomAffine(nn.Module):
    def __init__(self, degrees, translate=None, scale=None, shear=None, interpolation=InterpolationMode.NEAREST, fill=0,
                 center=None):
        super().__init__()
        self.degrees = _setup_angle(degrees, name='degrees', req_sizes=(2,))
        self.translate = translate
        self.scale = scale
        self.shear = _setup_angle(shear, name='shear', req_sizes=(2, 4)) if shear is not None else None
        self.interpolation = interpolation
        self.fill = fill
        self.center = center

    @staticmethod
    def get_params(degrees, translate, scale_ranges, shears, img_size):
        angle = float(torch.empty(1).uniform_(float(degrees[0]), float(degrees[1])).item())
        if translate is not None:
            max_dx = float(translate[0] * img_size[0])
            max_dy = float(translate[1] * img_size[1])
            tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
            ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
            translations = (tx, ty)
        else:
            translations = (0, 0)
        scale = float(torch.empty(1).uniform_(scale_ranges[0], scale_ranges[1]).item()) if scale_ranges is not None else 1.0
        shear_x = shear_y = 0.0
        if shears is not None:
            shear_x = float(torch.empty(1).uniform_(shears[0], shears[1]).item())
            if len(shears) == 4:
                shear_y = float(torch.empty(1).uniform_(shears[2], shears[3]).item())
        return angle, translations, scale, (shear_x, shear_y)

    def forward(self, img):
        fill = self.fill
        channels, height, width = F.get_dimensions(img)
        if isinstance(fill, (int, float)):
            fill = [float(fill)] * channels
        else:
            fill = [float(f) for f in fill]
        img_size = [width, height]
        ret = self.get_params(self.degrees, self.translate, self.scale, self.shear, img_size)
        return F.affine(img, *ret, interpolation=self.interpolation, fill=fill, center=self.center)
