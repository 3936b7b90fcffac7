"""Scaffolding stub (test infrastructure, never shipped to the product path): the torchvision.transforms.functional surface
the reference touches, restated for float tensors from the published torchvision 0.15.2 algorithms (requirements.txt:111;
torchvision itself is not installed in this image and not vendored in /root/reference).

crop/resize: tensor semantics of 0.15 (resize of a float tensor = F.interpolate, bilinear WITHOUT antialias,
align_corners=False).  Augmentation ops (transforms/_functional_tensor.py): _blend, rgb_to_grayscale, adjust_brightness /
contrast / saturation / hue (_rgb2hsv, _hsv2rgb), gaussian_blur (reflect padding, separable gaussian as one 2-D depthwise
conv), adjust_sharpness (3x3 smoothing kernel with centre 5, borders kept), affine (inverse matrix -> affine grid ->
grid_sample, align_corners=False, zero padding, `fill` through a mask channel)."""
import enum
import math

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


def get_dimensions(img):
    return [img.shape[-3], img.shape[-2], img.shape[-1]]


# ---------------------------------------------------------------------------------------- colour
def _blend(img1, img2, ratio):
    ratio = float(ratio)
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, 1.0).to(img1.dtype)


def rgb_to_grayscale(img, num_output_channels=1):
    r, g, b = img.unbind(dim=-3)
    l_img = (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype)
    return l_img.unsqueeze(dim=-3)


def adjust_brightness(img, brightness_factor):
    return _blend(img, torch.zeros_like(img), brightness_factor)


def adjust_contrast(img, contrast_factor):
    mean = torch.mean(rgb_to_grayscale(img).to(img.dtype), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, contrast_factor)


def adjust_saturation(img, saturation_factor):
    return _blend(img, rgb_to_grayscale(img), saturation_factor)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = hr + hg + hb
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - s * f)), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6, device=i.device).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum('...ijk, ...xijk -> ...xjk', mask.to(dtype=img.dtype), a4)


def adjust_hue(img, hue_factor):
    if not (-0.5 <= hue_factor <= 0.5):
        raise ValueError(f'hue_factor ({hue_factor}) is not in [-0.5, 0.5].')
    img = _rgb2hsv(img)
    h, s, v = img.unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    img = torch.stack((h, s, v), dim=-3)
    return _hsv2rgb(img)


# ---------------------------------------------------------------------------------------- blur / sharpen
def _get_gaussian_kernel1d(kernel_size, sigma):
    ksize_half = (kernel_size - 1) * 0.5
    x = torch.linspace(-ksize_half, ksize_half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img, kernel_size, sigma=None):
    if isinstance(kernel_size, int):
        kernel_size = [kernel_size, kernel_size]
    if isinstance(sigma, (int, float)):
        sigma = [float(sigma), float(sigma)]
    k1x = _get_gaussian_kernel1d(kernel_size[0], sigma[0]).to(img.dtype)
    k1y = _get_gaussian_kernel1d(kernel_size[1], sigma[1]).to(img.dtype)
    kernel = torch.mm(k1y[:, None], k1x[None, :])
    kernel = kernel.expand(img.shape[-3], 1, kernel.shape[0], kernel.shape[1])
    squeeze = img.dim() < 4
    x = img.unsqueeze(0) if squeeze else img
    padding = [kernel_size[0] // 2, kernel_size[0] // 2, kernel_size[1] // 2, kernel_size[1] // 2]
    x = F.pad(x, padding, mode='reflect')
    x = F.conv2d(x, kernel, groups=x.shape[-3])
    return x.squeeze(0) if squeeze else x


def _blurred_degenerate_image(img):
    kernel = torch.ones((3, 3), dtype=img.dtype, device=img.device)
    kernel[1, 1] = 5.0
    kernel /= kernel.sum()
    kernel = kernel.expand(img.shape[-3], 1, kernel.shape[0], kernel.shape[1])
    squeeze = img.dim() < 4
    x = img.unsqueeze(0) if squeeze else img
    result_tmp = F.conv2d(x, kernel, groups=x.shape[-3])
    if squeeze:
        result_tmp = result_tmp.squeeze(0)
    result = img.clone()
    result[..., 1:-1, 1:-1] = result_tmp
    return result


def adjust_sharpness(img, sharpness_factor):
    if img.size(-1) <= 2 or img.size(-2) <= 2:
        return img
    return _blend(img, _blurred_degenerate_image(img), sharpness_factor)


# ---------------------------------------------------------------------------------------- affine
def _get_inverse_affine_matrix(center, angle, translate, scale, shear, inverted=True):
    rot = math.radians(angle)
    sx = math.radians(shear[0])
    sy = math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    assert inverted
    matrix = [d, -b, 0.0, -c, a, 0.0]
    matrix = [x / scale for x in matrix]
    matrix[2] += matrix[0] * (-cx - tx) + matrix[1] * (-cy - ty)
    matrix[5] += matrix[3] * (-cx - tx) + matrix[4] * (-cy - ty)
    matrix[2] += cx
    matrix[5] += cy
    return matrix


def _gen_affine_grid(theta, w, h, ow, oh):
    d = 0.5
    base_grid = torch.empty(1, oh, ow, 3, dtype=theta.dtype, device=theta.device)
    x_grid = torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow, device=theta.device)
    base_grid[..., 0].copy_(x_grid)
    y_grid = torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh, device=theta.device).unsqueeze_(-1)
    base_grid[..., 1].copy_(y_grid)
    base_grid[..., 2].fill_(1)
    rescaled_theta = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=theta.dtype, device=theta.device)
    output_grid = base_grid.view(1, oh * ow, 3).bmm(rescaled_theta)
    return output_grid.view(1, oh, ow, 2)


def affine(img, angle, translate, scale, shear, interpolation=InterpolationMode.NEAREST, fill=None, center=None):
    if isinstance(shear, (int, float)):
        shear = [shear, 0.0]
    shear = [float(s) for s in shear]
    if len(shear) == 1:
        shear = [shear[0], shear[0]]
    translate_f = [1.0 * t for t in translate]
    matrix = _get_inverse_affine_matrix([0.0, 0.0], angle, translate_f, scale, shear)
    theta = torch.tensor(matrix, dtype=img.dtype, device=img.device).reshape(1, 2, 3)
    shape = img.shape
    grid = _gen_affine_grid(theta, w=shape[-1], h=shape[-2], ow=shape[-1], oh=shape[-2])
    squeeze = img.dim() < 4
    x = img.unsqueeze(0) if squeeze else img
    if x.shape[0] > 1:
        grid = grid.expand(x.shape[0], grid.shape[1], grid.shape[2], grid.shape[3])
    if fill is not None:
        mask = torch.ones((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
        x = torch.cat((x, mask), dim=1)
    mode = interpolation.value if isinstance(interpolation, InterpolationMode) else interpolation
    x = F.grid_sample(x, grid, mode=mode, padding_mode='zeros', align_corners=False)
    if fill is not None:
        mask = x[:, -1:, :, :]
        x = x[:, :-1, :, :]
        mask = mask.expand_as(x)
        fill_list = fill if isinstance(fill, (tuple, list)) else [float(fill)]
        fill_img = torch.tensor(fill_list, dtype=x.dtype, device=x.device).view(1, len(fill_list), 1, 1).expand_as(x)
        if mode == 'nearest':
            mask = mask < 0.5
            x[mask] = fill_img[mask]
        else:
            x = x * mask + (1.0 - mask) * fill_img
    return x.squeeze(0) if squeeze else x
