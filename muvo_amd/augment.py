"""Host side of the training-time augmentation (muvo/models/preprocess.py:295-367): the random draws.

The reference decides per frame / per sample on the HOST with torch's CPU generator (`torch.rand(1)`,
`torch.empty(1).uniform_()`, `torch.randperm(4)`, `torch.randint`), inside PixelAugmentation.forward, torchvision's
RandomApply / ColorJitter.get_params, RouteAugmentation.forward and RandomAffine.get_params (torchvision 0.15.2).  This module
makes exactly the same calls in exactly the same order, so with the same `torch.manual_seed` the draws ARE the reference's, and
packs them into the two small tables the kernels read (csrc/augment.hip).  No arithmetic on images happens here."""
import math

import torch

from muvo_amd.ops import PIXAUG_STRIDE, ROUTEAUG_STRIDE


def _jitter_range(value, center=1.0, clip_first_on_zero=True):
    """torchvision ColorJitter._check_input for a scalar setting."""
    value = [center - float(value), center + float(value)]
    if clip_first_on_zero:
        value[0] = max(value[0], 0.0)
    return None if value[0] == value[1] == center else value


def inverse_affine_matrix(angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix(center=(0, 0), ...) in Python floats."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return m


def _affine_params(degrees, translate, scale_ranges, shears, img_size):
    """RandomAffine.get_params: the RNG calls and their order."""
    angle = float(torch.empty(1).uniform_(float(degrees[0]), float(degrees[1])).item())
    max_dx, max_dy = float(translate[0] * img_size[0]), float(translate[1] * img_size[1])
    tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
    ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
    scale = float(torch.empty(1).uniform_(scale_ranges[0], scale_ranges[1]).item())
    shear_x = float(torch.empty(1).uniform_(shears[0], shears[1]).item())
    shear_y = float(torch.empty(1).uniform_(shears[2], shears[3]).item()) if len(shears) == 4 else 0.0
    return angle, (tx, ty), scale, (shear_x, shear_y)


def draw_pixel_params(cfg, b, s):
    """(b*s, PIXAUG_STRIDE) float32 CPU table; RNG calls as PixelAugmentation.forward (preprocess.py:316-331)."""
    a = cfg.IMAGE.AUGMENTATION
    ranges = [_jitter_range(a.COLOR_JITTER_BRIGHTNESS), _jitter_range(a.COLOR_JITTER_CONTRAST),
              _jitter_range(a.COLOR_JITTER_SATURATION), _jitter_range(a.COLOR_JITTER_HUE, center=0.0, clip_first_on_zero=False)]
    assert a.BLUR_PROB + a.SHARPEN_PROB <= 1 and int(a.BLUR_WINDOW) == 5, 'the blur kernel is the reference default 5x5'
    t = torch.zeros(b * s, PIXAUG_STRIDE)
    for f in range(b * s):
        rand_value = torch.rand(1)                       # compared as a float32 tensor, like the reference does
        if rand_value < a.BLUR_PROB:
            t[f, 0], t[f, 1] = 1, torch.empty(1).uniform_(a.BLUR_STD[0], a.BLUR_STD[1]).item()
        elif rand_value < a.BLUR_PROB + a.SHARPEN_PROB:
            t[f, 0], t[f, 1] = 2, torch.empty(1).uniform_(a.SHARPEN_FACTOR[0], a.SHARPEN_FACTOR[1]).item()
        if a.COLOR_PROB < torch.rand(1):                 # transforms.RandomApply: skip
            continue
        t[f, 2] = 1
        fn_idx = torch.randperm(4)                       # ColorJitter.get_params
        factors = [None if r is None else float(torch.empty(1).uniform_(r[0], r[1])) for r in ranges]
        for k, op in enumerate(fn_idx.tolist()):
            t[f, 3 + k] = op if factors[op] is not None else 4
        for op in range(4):
            t[f, 7 + op] = 0.0 if factors[op] is None else factors[op]
    return t


def draw_route_params(cfg, b, size):
    """(b, ROUTEAUG_STRIDE) float32 CPU table; RNG calls as RouteAugmentation.forward (preprocess.py:349-363)."""
    r = cfg.ROUTE
    drop, eor, small, large = (r.AUGMENTATION_DROPOUT, r.AUGMENTATION_END_OF_ROUTE, r.AUGMENTATION_SMALL_ROTATION,
                               r.AUGMENTATION_LARGE_ROTATION)
    assert drop + eor + small + large <= 1
    shears = [float(v) for v in r.AUGMENTATION_SHEAR]
    t = torch.zeros(b, ROUTEAUG_STRIDE)
    for i in range(b):
        rand_value = torch.rand(1)
        if rand_value < drop:
            t[i, 0] = 1
        elif rand_value < drop + eor:
            t[i, 0], t[i, 1] = 2, int(torch.randint(size, (1,)))
        elif rand_value < drop + eor + small + large:
            deg = float(r.AUGMENTATION_DEGREES) if rand_value < drop + eor + small else 180.0
            angle, tr, sc, sh = _affine_params((-deg, deg), r.AUGMENTATION_TRANSLATE, r.AUGMENTATION_SCALE, shears, [size, size])
            t[i, 0] = 3
            t[i, 2:8] = torch.tensor(inverse_affine_matrix(angle, [1.0 * tr[0], 1.0 * tr[1]], sc, sh), dtype=torch.float32)
    return t
