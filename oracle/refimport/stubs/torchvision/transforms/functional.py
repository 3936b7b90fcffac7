"""Scaffolding stub: tensor crop/resize with torchvision-0.15 tensor semantics
(resize of a float tensor = F.interpolate, bilinear without antialias, align_corners=False)."""
import enum
import torch
import torch.nn.functional as F


class InterpolationMode(enum.Enum):
    NEAREST = 'nearest'
    BILINEAR = 'bilinear'


def crop(img, top, left, height, width):
    return img[..., top:top + height, left:left + width]


def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
    assert img.dim() == 4
    size = tuple(size)
    if tuple(img.shape[-2:]) == size:
        return img
    is_float = img.is_floating_point()
    x = img if is_float else img.float()
    if interpolation == InterpolationMode.NEAREST:
        y = F.interpolate(x, size=size, mode='nearest')
    else:
        y = F.interpolate(x, size=size, mode='bilinear', align_corners=False, antialias=bool(antialias))
    if not is_float:
        y = y.round().to(img.dtype)
    return y
