"""ORACLE — test infrastructure only.  Never imported by the product path (muvo_amd/*).

A plain PyTorch-CPU fp32 restatement of the reference training step, written from the
behaviour of the reference (not copied), each piece citing the reference file:line it
follows.  Pinned against the real reference imported in the build container
(oracle/refimport/make_golden.py -> tests/golden/*.json|npz); the reference itself has
no tests, so those fixtures are the only pin ("parity unpinned by the reference's own
tests").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Module/parameter names equal the reference's Mile.state_dict() (680 entries) so the
deterministic weights keyed by name load into both.
"""
import math
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
def base_1d_cfg() -> dict:
    """Effective test_base_1d.yml values that gate the hot path."""
    return dict(
        TRANSFORMER_CHANNELS=384, EMBEDDING_DIM=512, HIDDEN_STATE_DIM=1024, STATE_DIM=512,
        ACTION_LATENT_DIM=64, ACTION_DIM=2, ROUTE_CHANNELS=16, SPEED_CHANNELS=16, SPEED_NORM=5.0,
        VOXEL_DIMENSION=64, VOXEL_N_CLASSES=2, LIDAR_RE_CHANNELS=4, LIDAR_RE_SCALE=50.0,
        CROP=(64, 138, 896, 458), ROUTE_SIZE=64, MEAN=(0.485, 0.456, 0.406), STD=(0.229, 0.224, 0.225),
        W_ACTION=1.0, W_PROB=1e-3, KL_ALPHA=0.75, W_LIDAR_RE=0.1, W_VOXEL=0.1, W_RGB=0.1,
        USE_PRIOR_PROB=0.15, LR=1e-4, WEIGHT_DECAY=0.01,
    )


def bev_out_of_view_mask(fov, image_width, resolution, crop_left, crop_right, bev_w, bev_h, offset_forward, camera_forward):
    """(bev_h, bev_w) bool: True = this bird's-eye-view cell is outside the camera's horizontal field of view or behind the ego
    vehicle (EVAL.MASK_VIEW; muvo/utils/geometry_utils.py:37-61).  Column u of a ground point (x right, z forward) through a
    pinhole with the principal point moved by the crop: u = x / z * f + c_u; visible when 0 <= u < cropped width.  Rows run
    from the far end of the grid towards the vehicle; the rows between camera and grid end are all masked."""
    import numpy as np
    f = image_width / (2 * np.tan(fov * np.pi / 360.0))
    c_u = image_width / 2 - crop_left
    half = np.round((bev_w // 2) * resolution, decimals=1)
    cam_off = (bev_h / 2 + offset_forward) * resolution + camera_forward
    top = np.round(bev_h * resolution - cam_off, decimals=1)
    x, z = np.arange(-half, half, resolution), np.arange(0.01, top, resolution)
    u = x / z[:, None] * f + c_u
    visible = (u >= 0) & (u < crop_right - crop_left)
    behind = np.ones((int(cam_off / resolution), visible.shape[1]), dtype=bool)
    return np.vstack([~visible[::-1], behind])


# --------------------------------------------------------------------------- preprocess
def _tv_blend(a, b, ratio):
    """torchvision _blend for float images: clamp(ratio*a + (1-ratio)*b, 0, 1)."""
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def _tv_gray(img):
    return (0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2]).unsqueeze(0)


def _tv_hue(img, hf):
    """torchvision adjust_hue on a float (3,H,W) image: RGB -> HSV, h = (h + hf) mod 1, HSV -> RGB."""
    r, g, b = img[0], img[1], img[2]
    maxc, minc = img.max(0).values, img.min(0).values
    eqc = maxc == minc
    cr = maxc - minc
    one = torch.ones_like(maxc)
    sat = cr / torch.where(eqc, one, maxc)
    div = torch.where(eqc, one, cr)
    rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
    h = (maxc == r) * (bc - gc) + ((maxc == g) & (maxc != r)) * (2.0 + rc - bc) + ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod(h / 6.0 + 1.0, 1.0)
    h = (h + hf) % 1.0
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int64) % 6
    v = maxc
    p = (v * (1.0 - sat)).clamp(0, 1)
    q = (v * (1.0 - sat * f)).clamp(0, 1)
    t = (v * (1.0 - sat * (1.0 - f))).clamp(0, 1)
    table = torch.stack([torch.stack(c) for c in ((v, q, p, p, t, v), (t, v, v, q, p, p), (p, p, t, v, v, q))])  # (3, 6, H, W)
    return table.gather(1, i.expand(3, 1, *i.shape)).squeeze(1)


def pixel_augmentation(img: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
    """PixelAugmentation.forward (preprocess.py:316-333) with the random draws as an explicit table: img (b,s,3,H,W) in
    [0,1] is modified IN PLACE and returned; params (b*s, 16): [0] 0 none / 1 gaussian blur 5x5 / 2 sharpen, [1] sigma |
    factor, [2] colour jitter on, [3..6] op order (0 brightness 1 contrast 2 saturation 3 hue, 4 = op switched off by the
    config), [7..10] factors.  Image algorithms: torchvision 0.15.2 (gaussian_blur: reflect padding + 2-D gaussian;
    adjust_sharpness: blend with the 3x3 [1 1 1; 1 5 1; 1 1 1]/13 smoothing, border pixels kept; ColorJitter ops)."""
    b, s = img.shape[:2]
    for f in range(b * s):
        P = params[f]
        x = img[f // s, f % s]
        mode = int(P[0])
        if mode == 1:
            t = torch.linspace(-2.0, 2.0, steps=5)
            k = torch.exp(-0.5 * (t / float(P[1])).pow(2))
            k = k / k.sum()
            k2 = (k[:, None] * k[None, :]).expand(3, 1, 5, 5)
            x = F.conv2d(F.pad(x[None], (2, 2, 2, 2), mode='reflect'), k2, groups=3)[0]
        elif mode == 2 and x.shape[-1] > 2 and x.shape[-2] > 2:
            k = torch.ones(3, 3)
            k[1, 1] = 5.0
            k = (k / k.sum()).expand(3, 1, 3, 3)
            smooth = x.clone()
            smooth[:, 1:-1, 1:-1] = F.conv2d(x[None], k, groups=3)[0]
            x = _tv_blend(x, smooth, float(P[1]))
        if int(P[2]):
            for kk in range(4):
                op = int(P[3 + kk])
                if op == 0:
                    x = _tv_blend(x, torch.zeros_like(x), float(P[7]))
                elif op == 1:
                    x = _tv_blend(x, _tv_gray(x).mean(), float(P[8]))
                elif op == 2:
                    x = _tv_blend(x, _tv_gray(x), float(P[9]))
                elif op == 3:
                    x = _tv_hue(x, float(P[10]))
        img[f // s, f % s] = x
    return img


def route_augmentation(route: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
    """RouteAugmentation.forward (preprocess.py:349-365) with explicit draws: route (b,s,3,H,W) in [0,1]; params (b, 8):
    [0] 0 none / 1 drop / 2 end of route / 3 affine, [1] leading rows zeroed, [2..7] torchvision's inverse affine matrix.
    Affine = torchvision F.affine(nearest, fill 0): affine grid in normalised coordinates + grid_sample(align_corners=False)."""
    out = route.clone()
    b, s, _, H, W = route.shape
    for i in range(b):
        mode = int(params[i, 0])
        if mode == 1:
            out[i] = 0
        elif mode == 2:
            out[i][:, :, :int(params[i, 1])] = 0
        elif mode == 3:
            theta = params[i, 2:8].reshape(1, 2, 3).to(route.dtype)
            base = torch.empty(1, H, W, 3)
            base[..., 0] = torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W)
            base[..., 1] = torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H).unsqueeze(-1)
            base[..., 2] = 1
            grid = base.view(1, H * W, 3).bmm(theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H])).view(1, H, W, 2)
            out[i] = F.grid_sample(route[i], grid.expand(s, H, W, 2), mode='nearest', padding_mode='zeros', align_corners=False)
    return out


def _aa_weights(in_size: int, out_size: int):
    """Per output index: (first input index, normalised triangle-filter weights) of the antialiased linear resize -
    torchvision 0.15.2 `resize(tensor, size, antialias=True)` = torch `F.interpolate(mode='bilinear', antialias=True,
    align_corners=False)` = ATen `_compute_indices_min_size_weights_aa` with the linear filter (third-party: torch 2.0,
    aten/src/ATen/native/cpu/UpSampleKernel.cpp): scale = in / out; support = scale when down-scaling, else 1; centre =
    scale * (i + 0.5); taps [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image; weight of tap j =
    max(0, 1 - |(j - centre + 0.5) / max(scale, 1)|), normalised to sum 1.  float32 arithmetic."""
    scale = torch.tensor(in_size / out_size, dtype=torch.float32)
    support = scale if scale >= 1.0 else torch.tensor(1.0)
    inv = 1.0 / scale if scale >= 1.0 else torch.tensor(1.0)
    rows = []
    for i in range(out_size):
        center = scale * (i + 0.5)
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        j = torch.arange(lo, hi, dtype=torch.float32)
        w = (1.0 - ((j - center + 0.5) * inv).abs()).clamp(min=0.0)
        rows.append((lo, (w / w.sum()).float()))
    return rows


def resize_bilinear_aa(x: torch.Tensor, size) -> torch.Tensor:
    """(N, C, H, W) float -> (N, C, h, w): horizontal pass first, then vertical (the order of ATen's separable kernel)."""
    h, w = size
    wx, wy = _aa_weights(x.shape[-1], w), _aa_weights(x.shape[-2], h)
    tmp = torch.stack([(x[..., lo:lo + len(k)] * k).sum(-1) for lo, k in wx], -1)
    return torch.stack([(tmp[..., lo:lo + len(k), :] * k[:, None]).sum(-2) for lo, k in wy], -2)


def preprocess(batch: Dict[str, torch.Tensor], cfg: dict, pixel_aug=None, route_aug=None) -> Dict[str, torch.Tensor]:
    """muvo/models/preprocess.py:201-225 (+ prepare_bev_labels :102-186).  pixel_aug / route_aug: the training-time
    augmentation (preprocess.py:213-214) with its random draws as explicit tables (None = off); it runs after the label
    pyramids were made and alters `rgb_label_1` (which IS the image) but not `rgb_label_2/4`."""
    out = dict(batch)
    img = batch['image'].float() / 255
    route = batch['route_map'].float() / 255
    b, s = img.shape[:2]
    rs = cfg['ROUTE_SIZE']
    if tuple(route.shape[-2:]) != (rs, rs):  # functional_resize default = NEAREST (preprocess.py:207,277)
        route = F.interpolate(route.flatten(0, 1), size=(rs, rs), mode='nearest').view(b, s, 3, rs, rs)
    left, top, right, bottom = cfg['CROP']
    img = img[..., top:bottom, left:right]
    intr = batch['intrinsics'].clone()
    intr[..., 0, 2] -= left
    intr[..., 1, 2] -= top
    if cfg.get('EVAL_RESOLUTION_FACTOR', 1) != 1:       # EVAL.RESOLUTION (preprocess.py:209-210, functional_resize_batch :252-273)
        sc = 1 / cfg['EVAL_RESOLUTION_FACTOR']
        h1, w1 = int(round(img.shape[-2] * sc)), int(round(img.shape[-1] * sc))
        img = resize_bilinear_aa(img.flatten(0, 1), (h1, w1)).view(b, s, 3, h1, w1)
        intr[..., :2, :] *= sc
    out['intrinsics'] = intr
    # rgb labels: bilinear (no antialias), each level from the previous (preprocess.py:104-113)
    out['rgb_label_1'] = img
    h, w = img.shape[-2:]
    prev = img
    for f in (2, 4):
        prev = F.interpolate(prev.flatten(0, 1), size=(h // f, w // f), mode='bilinear',
                             align_corners=False).view(b, s, 3, h // f, w // f)
        out[f'rgb_label_{f}'] = prev
    # LOSSES.RGB_INSTANCE (preprocess.py:115-125,242-243): the instance mask is cropped like the image, nearest pyramid
    if 'image_instance_mask' in batch:
        im = batch['image_instance_mask'][..., top:bottom, left:right]
        out['image_instance_mask'] = out['image_instance_mask_1'] = im
        h, w = im.shape[-2:]
        prev = im
        for f in (2, 4):
            prev = F.interpolate(prev.flatten(0, 1).float(), size=(h // f, w // f), mode='nearest').to(im.dtype).view(b, s, 1, h // f, w // f)
            out[f'image_instance_mask_{f}'] = prev
    # range view (preprocess.py:150-162)
    rv = batch['range_view_pcd_xyzd'].float() / cfg['LIDAR_RE_SCALE']
    out['range_view_pcd_xyzd'] = rv
    out['range_view_label_1'] = rv
    h, w = rv.shape[-2:]
    prev = rv
    for f in (2, 4):
        prev = F.interpolate(prev.flatten(0, 1), size=(h // f, w // f), mode='nearest').view(b, s, -1, h // f, w // f)
        out[f'range_view_label_{f}'] = prev
    # voxel labels: nearest on uint8 (preprocess.py:176-186)
    vox = batch['voxel']
    out['voxel_label_1'] = vox
    x, y, z = vox.shape[-3:]
    prev = vox
    for f in (2, 4):
        prev = F.interpolate(prev.flatten(0, 1), size=(x // f, y // f, z // f), mode='nearest').view(
            b, s, 1, x // f, y // f, z // f)
        out[f'voxel_label_{f}'] = prev
    # bird's-eye-view labels, when present (preprocess.py:50-100, EVAL.MASK_VIEW off)
    view_mask = None
    if cfg.get('MASK_VIEW'):         # EVAL.MASK_VIEW (preprocess.py:20-21,52-54,70-72) with the default geometry (config.py:111-141)
        view_mask = torch.from_numpy(bev_out_of_view_mask(*cfg.get('MASK_VIEW_GEOMETRY', (100, 960, 0.2, 64, 896, 192, 192, -64, 1.0))))
    if 'birdview_label' in batch:
        if view_mask is not None:
            batch['birdview_label'][:, :, :, view_mask] = 0
        bev = torch.rot90(batch['birdview_label'], k=-1, dims=[3, 4]).contiguous()
        out['birdview_label'] = out['birdview_label_1'] = bev
        h, w = bev.shape[-2:]
        prev = bev
        for f in (2, 4):
            prev = F.interpolate(prev.flatten(0, 1).float(), size=(h // f, w // f), mode='nearest').to(bev.dtype).view(b, s, 1, h // f, w // f)
            out[f'birdview_label_{f}'] = prev
    if 'instance_label' in batch:
        if view_mask is not None:
            batch['instance_label'][:, :, :, view_mask] = 0
        inst = torch.rot90(batch['instance_label'], k=-1, dims=[3, 4]).contiguous()
        out['instance_label'] = out['instance_label_1'] = inst
        out['center_label_1'], out['offset_label_1'] = instance_center_offset(inst, 255, 4.0)    # config.py:234-235
        h, w = inst.shape[-2:]
        prev = inst
        for f in (2, 4):
            prev = F.interpolate(prev.flatten(0, 1).float(), size=(h // f, w // f), mode='nearest').to(inst.dtype).view(b, s, 1, h // f, w // f)
            out[f'instance_label_{f}'] = prev
            out[f'center_label_{f}'], out[f'offset_label_{f}'] = instance_center_offset(prev, 255, 4.0 / f)
    # inputs of the config-off heads, when present (preprocess.py:127-149,164-175,228-241): crop like the image, pyramids
    if 'semantic_image' in batch:
        sem = batch['semantic_image'][..., top:bottom, left:right]
        out['semantic_image'] = sem
        out['semantic_image_label_1'] = sem
        h, w = sem.shape[-2:]
        prev = sem
        for f in (2, 4):
            prev = F.interpolate(prev.flatten(0, 1).float(), size=(h // f, w // f), mode='nearest').to(sem.dtype).view(b, s, 1, h // f, w // f)
            out[f'semantic_image_label_{f}'] = prev
    if 'depth' in batch:
        dep = batch['depth'][..., top:bottom, left:right].float()
        out['depth'] = dep
        out['depth_label_1'] = dep
        h, w = dep.shape[-2:]
        prev = dep
        for f in (2, 4):
            prev = F.interpolate(prev.flatten(0, 1), size=(h // f, w // f), mode='bilinear', align_corners=False).view(b, s, 1, h // f, w // f)
            out[f'depth_label_{f}'] = prev
    if 'range_view_pcd_seg' in batch:
        seg = batch['range_view_pcd_seg']
        out['range_view_seg_label_1'] = seg
        h, w = seg.shape[-2:]
        prev = seg
        for f in (2, 4):
            prev = F.interpolate(prev.flatten(0, 1).float(), size=(h // f, w // f), mode='nearest').to(seg.dtype).view(b, s, 1, h // f, w // f)
            out[f'range_view_seg_label_{f}'] = prev
    if pixel_aug is not None:
        img = img.contiguous()
        out['rgb_label_1'] = img                 # stays the same tensor as the (augmented) image
        pixel_augmentation(img, pixel_aug)
    if route_aug is not None:
        route = route_augmentation(route, route_aug)
    mean = torch.tensor(cfg['MEAN']).view(3, 1, 1)
    std = torch.tensor(cfg['STD']).view(3, 1, 1)
    out['image'] = (img - mean) / std
    out['route_map'] = (route - mean) / std
    return out


# --------------------------------------------------------------------------- building blocks
class ResBlock(nn.Module):
    """timm BasicBlock / muvo/layers/layers.py:9-66."""

    def __init__(self, cin, cout, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, 2, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        sc = x if self.downsample is None else self.downsample(x)
        return F.relu(y + sc)


class ResNet18(nn.Module):
    """timm resnet18 features_only (mile.py:24-26,81-83; common.py:15)."""

    def __init__(self, in_chans=3, out_indices=(2, 3, 4)):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, c in enumerate((64, 128, 256, 512)):
            st = 1 if i == 0 else 2
            setattr(self, f'layer{i + 1}', nn.Sequential(ResBlock(cin, c, st, st != 1), ResBlock(c, c)))
            cin = c
        self.out_indices = out_indices

    def forward(self, x):
        feats = []
        x = F.relu(self.bn1(self.conv1(x)))
        feats.append(x)
        x = F.max_pool2d(x, 3, 2, 1)
        for i in range(4):
            x = getattr(self, f'layer{i + 1}')(x)
            feats.append(x)
        return [feats[i] for i in self.out_indices]


def _cbr(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


class DecoderDS(nn.Module):
    """common.py:102-130."""

    def __init__(self, chans, cout):
        super().__init__()
        self.conv1 = _cbr(chans[0], cout)
        self.downsample_skip_convs = nn.ModuleList(_cbr(c, cout) for c in chans[1:])

    def forward(self, xs):
        x = self.conv1(xs[0])
        for i, conv in enumerate(self.downsample_skip_convs):
            stride = xs[i].shape[-1] // xs[i + 1].shape[-1]
            x = conv(xs[i + 1]) + F.max_pool2d(x, stride)
        return x


def position_embedding_sine(h, w, num_pos_feats, temperature=10000.0):
    """common.py:636-678 with normalize=True, scale=2*pi. Returns (1, 2*num_pos_feats, h, w)."""
    ones = torch.ones((1, h, w))
    y_embed = ones.cumsum(1, dtype=torch.float32)
    x_embed = ones.cumsum(2, dtype=torch.float32)
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * (2 * math.pi)
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class RouteEncode(nn.Module):
    """common.py:12-23."""

    def __init__(self, cout):
        super().__init__()
        self.backbone = ResNet18(3, (4,))
        self.fc = nn.Linear(512, cout)

    def forward(self, x):
        x = self.backbone(x)[0]
        return self.fc(x.mean(dim=(-1, -2)))


class FeatureConv(nn.Sequential):
    """mile.py:104-115 (BasicBlock s2 + BasicBlock + global avg pool + flatten)."""

    def __init__(self, cin, cout):
        super().__init__(ResBlock(cin, cout, 2, True), ResBlock(cout, cout), nn.AdaptiveAvgPool2d(1), nn.Flatten(1))


class Representation(nn.Module):
    """transition.py:5-25; nn.LeakyReLU(True) has slope 1.0 == identity."""

    def __init__(self, cin, latent):
        super().__init__()
        self.latent = latent
        self.module = nn.Sequential(nn.Linear(cin, cin), nn.Identity(), nn.Linear(cin, 2 * latent))

    def forward(self, x):
        mu, ls = torch.split(self.module(x), self.latent, dim=-1)
        return mu, 2 * torch.sigmoid(ls / 2) + 0.1


class RSSM(nn.Module):
    """transition.py:28-173, with the RNG made explicit (noise (b,s,2,S), use_prior flags)."""

    def __init__(self, emb, act, hid, st, alat):
        super().__init__()
        self.hid, self.st = hid, st
        self.pre_gru_net = nn.Sequential(nn.Linear(st, hid), nn.Identity())
        self.recurrent_model = nn.GRUCell(hid, hid)
        self.posterior_action_module = nn.Sequential(nn.Linear(act, alat), nn.Identity())
        self.posterior = Representation(hid + emb + alat, st)
        self.prior_action_module = nn.Sequential(nn.Linear(act, alat), nn.Identity())
        self.prior = Representation(hid + alat, st)

    def forward(self, emb, action, noise, use_prior):
        b, s, _ = emb.shape
        h = emb.new_zeros(b, self.hid)
        z = emb.new_zeros(b, self.st)
        keys = ('hidden_state', 'sample', 'mu', 'sigma')
        pri = {k: [] for k in keys}
        pos = {k: [] for k in keys}
        for t in range(s):
            a = torch.zeros_like(action[:, 0]) if t == 0 else action[:, t - 1]
            h = self.recurrent_model(self.pre_gru_net(z), h)
            pm, ps = self.prior(torch.cat([h, self.prior_action_module(a)], -1))
            pz = pm + ps * noise[:, t, 0]
            qm, qs = self.posterior(torch.cat([h, emb[:, t], self.posterior_action_module(a)], -1))
            qz = qm + qs * noise[:, t, 1]
            for d, vals in ((pri, (h, pz, pm, ps)), (pos, (h, qz, qm, qs))):
                for k, v in zip(keys, vals):
                    d[k].append(v)
            z = pz if use_prior[t] else qz
        return ({k: torch.stack(v, 1) for k, v in pri.items()}, {k: torch.stack(v, 1) for k, v in pos.items()})

    def imagine_step(self, h, z, a, eps):
        """transition.py:151-173 (prior roll-out step; eps = the explicit N(0,1) draw of sample_from_distribution)."""
        h = self.recurrent_model(self.pre_gru_net(z), h)
        pm, ps = self.prior(torch.cat([h, self.prior_action_module(a)], -1))
        return h, pm + ps * eps, pm, ps


def rssm_observe_step(rssm, h, z, a, emb, eps_prior, eps_post):
    """RSSM.observe_step (transition.py:130-149): the imagine step, then the posterior from (h, embedding, action latent).
    eps = 0 reproduces use_sample=False (the sample is the mean).  Returns (h, prior sample, posterior sample, mu_q, sigma_q)."""
    h, zp, _, _ = rssm.imagine_step(h, z, a, eps_prior)
    qm, qs = rssm.posterior(torch.cat([h, emb, rssm.posterior_action_module(a)], -1))
    return h, zp, qm + qs * eps_post, qm, qs


class SimState:
    """the latent memory deployment_forward / sim_forward keep between calls (mile.py:399-402)"""

    def __init__(self):
        self.last_h = self.last_sample = self.last_action = None
        self.count = 0


def sim_forward(model, state: SimState, batch, is_dreaming, receptive_field, stride_frames):
    """Mile.sim_forward (mile.py:925-1032) on a preprocessed batch, use_sample=False everywhere except the imagination, whose
    noise is zero here too (deterministic restatement: the reference draws it).  Every `stride_frames` calls (int(CARLA_FPS *
    DATASET.STRIDE_SEC), constants.py:3) the newest frame is encoded and the latent state advanced with the PREVIOUS call's
    action; the calls in between only count down.  Returns (output, output_imagine)."""
    b = batch['image'].shape[0]
    if state.count == 0:
        cut = {k: v[:, receptive_field - 1:].contiguous() for k, v in batch.items()}      # remove_past
        action_t = torch.cat([cut['throttle_brake'][:, 0], cut['steering'][:, 0]], -1)
        emb = model.encode({k: v[:, :1] for k, v in cut.items()})[:, -1]
        a_last = torch.zeros_like(action_t) if state.last_action is None else state.last_action
        H, S = model.cfg['HIDDEN_STATE_DIM'], model.cfg['STATE_DIM']
        h = action_t.new_zeros(b, H) if state.last_h is None else state.last_h
        z = action_t.new_zeros(b, S) if state.last_h is None else state.last_sample
        zero = torch.zeros(b, S)
        if is_dreaming:
            h, z, _, _ = model.rssm.imagine_step(h, z, a_last, zero)
        else:
            h, _, z, _, _ = rssm_observe_step(model.rssm, h, z, a_last, emb, zero, zero)
        state.last_h, state.last_sample, state.last_action = h, z, action_t
        state.count = stride_frames - 1
        batch = cut
    else:
        state.count -= 1
    st = torch.cat([state.last_h, state.last_sample], -1)
    pol = model.policy(st)
    out = {'throttle_brake': pol[:, :1].view(b, 1, 1), 'steering': pol[:, 1:].view(b, 1, 1), 'hidden_state': state.last_h,
           'sample': state.last_sample}
    for dec in model.main_decoders() + model.aux_decoders():
        for k, v in dec(st).items():
            out[k] = v.view(b, 1, *v.shape[1:])
    fh = batch['image'].shape[1] - 1
    imag = imagine(model, {'hidden_state': state.last_h, 'sample': state.last_sample, 'throttle_brake': batch['throttle_brake'],
                           'steering': batch['steering']}, fh, torch.zeros(b, max(fh, 1), model.cfg['STATE_DIM'])) if fh > 0 else {}
    return out, imag


class Policy(nn.Module):
    """common.py:53-68."""

    def __init__(self, c):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(c, c), nn.ReLU(), nn.Linear(c, c), nn.ReLU(), nn.Linear(c, c // 2),
                                nn.ReLU(), nn.Linear(c // 2, 2), nn.Tanh())

    def forward(self, x):
        return self.fc(x)


class _Head(nn.Module):
    def __init__(self, attr, conv):
        super().__init__()
        setattr(self, attr, nn.Sequential(conv))
        self._attr = attr

    def forward(self, x):
        return getattr(self, self._attr)(x)


class ConvDecoder(nn.Module):
    """common.py:549-632."""

    def __init__(self, latent, cout, const_size, head_attr, out_key):
        super().__init__()
        c = 512
        self.out_key = out_key
        self.linear = nn.Sequential(nn.Linear(latent, c), nn.Unflatten(-1, (c, 1, 1)))
        self.pre_transpose_conv = nn.Sequential(
            nn.ConvTranspose2d(c, c, const_size), nn.ELU(),
            nn.ConvTranspose2d(c, c, 5, 2, 2, 1), nn.ELU(),
            nn.ConvTranspose2d(c, c, 5, 2, 2, 1), nn.ELU(),
            nn.ConvTranspose2d(c, c, 6, 2, 2), nn.ELU())
        self.trans_conv1 = nn.Sequential(nn.ConvTranspose2d(c, 256, 6, 2, 2), nn.ELU())
        self.head_4 = _Head(head_attr, nn.Conv2d(256, cout, 1))
        self.trans_conv2 = nn.Sequential(nn.ConvTranspose2d(256, 128, 6, 2, 2), nn.ELU())
        self.head_2 = _Head(head_attr, nn.Conv2d(128, cout, 1))
        self.trans_conv3 = nn.Sequential(nn.ConvTranspose2d(128, 64, 6, 2, 2), nn.ELU())
        self.head_1 = _Head(head_attr, nn.Conv2d(64, cout, 1))

    def forward(self, x):
        x = self.pre_transpose_conv(self.linear(x))
        x = self.trans_conv1(x)
        o4 = self.head_4(x)
        x = self.trans_conv2(x)
        o2 = self.head_2(x)
        x = self.trans_conv3(x)
        o1 = self.head_1(x)
        return {f'{self.out_key}_4': o4, f'{self.out_key}_2': o2, f'{self.out_key}_1': o1}


class AdaIN3d(nn.Module):
    """common.py:227-246."""

    def __init__(self, latent, c):
        super().__init__()
        self.c = c
        self.latent_affine = nn.Linear(latent, 2 * c)

    def forward(self, x, w):
        mean = x.mean(dim=(-1, -2, -3), keepdim=True)
        x = x - mean
        std = torch.sqrt(torch.mean(x ** 2, dim=(-1, -2, -3), keepdim=True) + 1e-8)
        x = x / std
        style = self.latent_affine(w)[:, :, None, None, None]
        scale, bias = torch.split(style, self.c, dim=1)
        return scale * x + bias


class ConvIN3d(nn.Module):
    """common.py:190-202."""

    def __init__(self, cin, cout, latent):
        super().__init__()
        self.conv_act = nn.Sequential(nn.Conv3d(cin, cout, 3, 1, 1), nn.LeakyReLU(0.2))
        self.adaptive_norm = AdaIN3d(latent, cout)

    def forward(self, x, w):
        return self.adaptive_norm(self.conv_act(x), w)


class DecBlock3d(nn.Module):
    """common.py:161-172 (upsample=True)."""

    def __init__(self, cin, cout, latent):
        super().__init__()
        self.conv1 = ConvIN3d(cin, cout, latent)
        self.conv2 = ConvIN3d(cout, cout, latent)

    def forward(self, x, w):
        x = F.interpolate(x, scale_factor=2.0, mode='trilinear', align_corners=False)
        return self.conv2(self.conv1(x, w), w)


class AdaIN2d(nn.Module):
    """common.py:205-224."""

    def __init__(self, latent, c):
        super().__init__()
        self.c = c
        self.latent_affine = nn.Linear(latent, 2 * c)

    def forward(self, x, w):
        mean = x.mean(dim=(-1, -2), keepdim=True)
        x = x - mean
        std = torch.sqrt(torch.mean(x ** 2, dim=(-1, -2), keepdim=True) + 1e-8)
        x = x / std
        scale, bias = torch.split(self.latent_affine(w)[:, :, None, None], self.c, dim=1)
        return scale * x + bias


class ConvIN2d(nn.Module):
    """common.py:175-187."""

    def __init__(self, cin, cout, latent):
        super().__init__()
        self.conv_act = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.LeakyReLU(0.2))
        self.adaptive_norm = AdaIN2d(latent, cout)

    def forward(self, x, w):
        return self.adaptive_norm(self.conv_act(x), w)


class DecBlock2d(nn.Module):
    """common.py:147-159 (upsample=True)."""

    def __init__(self, cin, cout, latent):
        super().__init__()
        self.conv1 = ConvIN2d(cin, cout, latent)
        self.conv2 = ConvIN2d(cout, cout, latent)

    def forward(self, x, w):
        x = F.interpolate(x, scale_factor=2.0, mode='bilinear', align_corners=False)
        return self.conv2(self.conv1(x, w), w)


class SegmentationHead(nn.Module):
    """common.py:249-271."""

    def __init__(self, cin, n_classes, f):
        super().__init__()
        self.f = f
        self.segmentation_head = nn.Sequential(nn.Conv2d(cin, n_classes, 1))
        self.instance_offset_head = nn.Sequential(nn.Conv2d(cin, 2, 1))
        self.instance_center_head = nn.Sequential(nn.Conv2d(cin, 1, 1), nn.Sigmoid())

    def forward(self, x):
        return {f'bev_segmentation_{self.f}': self.segmentation_head(x), f'bev_instance_offset_{self.f}': self.instance_offset_head(x),
                f'bev_instance_center_{self.f}': self.instance_center_head(x)}


class BevDecoder(nn.Module):
    """common.py:370-424 (head='bev')."""

    def __init__(self, latent, n_classes, const_size=(3, 3)):
        super().__init__()
        n = 512
        self.constant_tensor = nn.Parameter(torch.randn(n, *const_size))
        self.first_norm = AdaIN2d(latent, n)
        self.first_conv = ConvIN2d(n, n, latent)
        self.middle_conv = nn.ModuleList(DecBlock2d(n, n, latent) for _ in range(3))
        self.conv1 = DecBlock2d(n, 256, latent)
        self.head_4 = SegmentationHead(256, n_classes, 4)
        self.conv2 = DecBlock2d(256, 128, latent)
        self.head_2 = SegmentationHead(128, n_classes, 2)
        self.conv3 = DecBlock2d(128, 64, latent)
        self.head_1 = SegmentationHead(64, n_classes, 1)

    def forward(self, w):
        x = self.constant_tensor.unsqueeze(0).repeat([w.shape[0], 1, 1, 1])
        x = self.first_conv(self.first_norm(x, w), w)
        for m in self.middle_conv:
            x = m(x, w)
        x = self.conv1(x, w)
        o4 = self.head_4(x)
        x = self.conv2(x, w)
        o2 = self.head_2(x)
        x = self.conv3(x, w)
        return {**o4, **o2, **self.head_1(x)}


def instance_center_offset(instance_label, ignore_index=255, sigma=3.0):
    """convert_instance_mask_to_center_and_offset_label (instance_utils.py:4-35): per frame and instance id the rounded
    centroid; centre = max over instances of exp(-d^2 / sigma^2), offset = centroid - pixel on the instance, ignore elsewhere."""
    inst = instance_label.squeeze(2)
    b, s, h, w = inst.shape
    center = torch.zeros(b, s, 1, h, w)
    offset = ignore_index * torch.ones(b, s, 2, h, w)
    x, y = torch.meshgrid(torch.arange(h, dtype=torch.float), torch.arange(w, dtype=torch.float), indexing='ij')
    for bi in range(b):
        for t in range(s):
            for iid in torch.unique(inst[bi, t]).tolist():
                if iid == 0:
                    continue
                m = inst[bi, t] == iid
                xc, yc = x[m].mean().round(), y[m].mean().round()
                g = torch.exp(-((xc - x) ** 2 + (yc - y) ** 2) / sigma ** 2)
                center[bi, t, 0] = torch.maximum(center[bi, t, 0], g)
                offset[bi, t, 0][m] = (xc - x)[m]
                offset[bi, t, 1][m] = (yc - y)[m]
    return center, offset


SEMANTIC_SEG_WEIGHTS = (1.0, 1.0, 1.0, 2.0, 3.0, 1.0, 1.0, 1.0)      # constants.py:33 (is_bev=True, trainer.py:61-66)


class VoxelDecoder1(nn.Module):
    """common.py:498-546."""

    def __init__(self, latent, n_classes, fc, const_size=(3, 3, 1)):
        super().__init__()
        self.constant_tensor = nn.Parameter(torch.randn(2 * fc, *const_size))
        self.first_norm = AdaIN3d(latent, 2 * fc)
        self.first_conv = ConvIN3d(2 * fc, fc, latent)
        self.middle_conv = nn.ModuleList(DecBlock3d(fc, fc, latent) for _ in range(3))
        self.conv1 = DecBlock3d(fc, fc // 2, latent)
        self.head_4 = _Head('segmentation_head', nn.Conv3d(fc // 2, n_classes, 1))
        self.conv2 = DecBlock3d(fc // 2, fc // 4, latent)
        self.head_2 = _Head('segmentation_head', nn.Conv3d(fc // 4, n_classes, 1))
        self.conv3 = DecBlock3d(fc // 4, fc // 8, latent)
        self.head_1 = _Head('segmentation_head', nn.Conv3d(fc // 8, n_classes, 1))

    def forward(self, w):
        x = self.constant_tensor.unsqueeze(0).repeat(w.shape[0], 1, 1, 1, 1)
        x = self.first_conv(self.first_norm(x, w), w)
        for m in self.middle_conv:
            x = m(x, w)
        x = self.conv1(x, w)
        o4 = self.head_4(x)
        x = self.conv2(x, w)
        o2 = self.head_2(x)
        x = self.conv3(x, w)
        o1 = self.head_1(x)
        return {'voxel_4': o4, 'voxel_2': o2, 'voxel_1': o1}


class DecoderUp(nn.Module):
    """common.py:71-99 (`Decoder`): coarse-to-fine skip decoder with bilinear upsampling (align_corners=False)."""

    def __init__(self, chans, cout):
        super().__init__()
        self.conv1 = _cbr(chans[-1], cout)
        self.upsample_skip_convs = nn.ModuleList(_cbr(c, cout) for c in reversed(chans[:-1]))
        self.out_channels = cout

    def forward(self, xs):
        x = self.conv1(xs[-1])
        for i, conv in enumerate(self.upsample_skip_convs):
            size = xs[-(i + 2)].shape[-2:]
            x = conv(xs[-(i + 2)]) + F.interpolate(x, size=size, mode='bilinear', align_corners=False)
        return x


BEV_DEFAULTS = dict(SIZE=(192, 192), RESOLUTION=0.2, OFFSET_FORWARD=-64, FEATURE_DOWNSAMPLE=4, D_BOUND=(1.0, 38.0, 1.0),
                    SPARSE=True, SPARSE_COUNT=10)   # config.py:135-144


class MileRef(nn.Module):
    """muvo/models/mile.py:16-161,284-402 (construction), :404-593 (forward/encode), base_1d branch; bev=True adds the
    MODEL.TRANSFORMER.BEV branch (mile.py:33-59,506-524): Decoder instead of DecoderDS, mono depth head, frustum pooling
    and the two-conv BEV down-sampling."""

    def __init__(self, cfg: dict = None, bev: bool = False, aux_heads=()):
        super().__init__()
        cfg = cfg or base_1d_cfg()
        self.cfg = cfg
        self.bev = bev
        self.aux_heads = tuple(aux_heads)      # subset of ('lidar_seg', 'sem_image', 'depth'): mile.py:337-363
        tc, emb = cfg['TRANSFORMER_CHANNELS'], cfg['EMBEDDING_DIM']
        self.encoder = ResNet18(3)
        self.feat_decoder = DecoderUp((128, 256, 512), tc) if bev else DecoderDS((128, 256, 512), tc)
        if bev:
            bc = {**BEV_DEFAULTS, **cfg.get('BEV', {})}
            ds_ = bc['FEATURE_DOWNSAMPLE']
            self.bev_args = dict(size=(bc['SIZE'][0] // ds_, bc['SIZE'][1] // ds_), scale=bc['RESOLUTION'] * ds_,
                                 offsetx=bc['OFFSET_FORWARD'] / ds_, dbound=list(bc['D_BOUND']), downsample=8)
            self.sparse_depth, self.sparse_depth_count = bc['SPARSE'], bc['SPARSE_COUNT']
            self.frustum_pooling = nn.Module()          # state-dict compatibility: the one persistent buffer (:80)
            self.frustum_pooling.register_buffer(
                'bev_intrinsics', frustum_grid(self.bev_args['size'], self.bev_args['scale'], self.bev_args['offsetx'])[3])
            n_bins = len(torch.arange(*bc['D_BOUND']))
            self.depth_decoder = DecoderUp((128, 256, 512), tc)
            self.depth = nn.Conv2d(tc, n_bins, kernel_size=1)
            self.bev_down_sample_4 = nn.Sequential(nn.Conv2d(tc, 512, kernel_size=5, stride=2, padding=2), nn.ReLU(),
                                                   nn.Conv2d(512, tc, kernel_size=5, stride=2, padding=2))
        self.range_view_encoder = ResNet18(4)
        self.range_view_decoder = DecoderDS((128, 256, 512), tc)
        self.type_embedding = nn.Parameter(torch.zeros(1, 1, tc, 2))
        self.encoder_layer = nn.TransformerEncoderLayer(d_model=tc, nhead=8, dropout=0.1)  # registered, unused
        self.transformer_encoder = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=tc, nhead=8, dropout=0.1), num_layers=6, enable_nested_tensor=False)
        self.image_feature_conv = FeatureConv(tc, emb)
        self.lidar_feature_conv = FeatureConv(tc, emb)
        self.backbone_route = RouteEncode(cfg['ROUTE_CHANNELS'])
        sc = cfg['SPEED_CHANNELS']
        self.speed_enc = nn.Sequential(nn.Linear(1, sc), nn.ReLU(), nn.Linear(sc, sc), nn.ReLU())
        self.features_combine = nn.Linear(2 * emb + cfg['ROUTE_CHANNELS'] + sc, emb)
        self.rssm = RSSM(emb, cfg['ACTION_DIM'], cfg['HIDDEN_STATE_DIM'], cfg['STATE_DIM'], cfg['ACTION_LATENT_DIM'])
        sd = cfg['HIDDEN_STATE_DIM'] + cfg['STATE_DIM']
        self.policy = Policy(sd)
        # seed sizes: the reference's constants (mile.py:322-336,391-396) unless the MODEL.CONSTANT_SIZE extension of
        # muvo_amd/config.py is in use (cfg keys RGB_CONST / LIDAR_CONST / VOXEL_CONST; no reference counterpart: unpinned)
        cs_rgb, cs_lidar = tuple(cfg.get('RGB_CONST', (5, 13))), tuple(cfg.get('LIDAR_CONST', (1, 16)))
        cs_voxel = tuple(cfg.get('VOXEL_CONST', (3, 3, 1)))
        if cfg.get('RGB_SUPERVISION', True):     # EVAL.RGB_SUPERVISION (mile.py:315-321)
            self.rgb_decoder = ConvDecoder(sd, 3, cs_rgb, 'rgb_head', 'rgb')
        self.lidar_re = ConvDecoder(sd, cfg['LIDAR_RE_CHANNELS'], cs_lidar, 'lidar_re_head', 'lidar_reconstruction')
        self.voxel_decoder = VoxelDecoder1(sd, cfg['VOXEL_N_CLASSES'], cfg['VOXEL_DIMENSION'], cs_voxel)
        if 'bev' in self.aux_heads:             # SEMANTIC_SEG (mile.py:307-313)
            self.bev_decoder = BevDecoder(sd, 8)
        if 'lidar_seg' in self.aux_heads:
            self.lidar_segmentation = ConvDecoder(sd, 9, cs_lidar, 'seg_head', 'lidar_segmentation')
        if 'sem_image' in self.aux_heads:
            self.sem_image_decoder = ConvDecoder(sd, 9, cs_rgb, 'sem_head', 'semantic_image')
        if 'depth' in self.aux_heads:
            self.depth_image_decoder = ConvDecoder(sd, 1, cs_rgb, 'depth_head', 'depth')

    def set_dropout(self, p: float):
        for m in self.modules():
            if isinstance(m, nn.Dropout):
                m.p = p
            if isinstance(m, nn.MultiheadAttention):
                m.dropout = p

    def encode(self, batch):
        b, s = batch['image'].shape[:2]
        image = batch['image'].flatten(0, 1)
        xs = self.encoder(image)
        x = self.feat_decoder(xs)
        if self.bev:                                                               # mile.py:506-524
            depth = self.depth(self.depth_decoder(xs)).softmax(dim=1)
            if self.sparse_depth:
                mask = torch.zeros(depth.shape, dtype=torch.bool)
                mask.scatter_(1, depth.topk(self.sparse_depth_count, dim=1)[1], 1)
            else:
                mask = torch.zeros(0)
            x = frustum_pool(x, depth, mask, batch['intrinsics'].flatten(0, 1), batch['extrinsics'].flatten(0, 1), **self.bev_args)
            x = self.bev_down_sample_4(x)
        lf = self.range_view_decoder(self.range_view_encoder(batch['range_view_pcd_xyzd'].flatten(0, 1)))
        nf = self.cfg['TRANSFORMER_CHANNELS'] // 2
        it = x + position_embedding_sine(x.shape[2], x.shape[3], nf)
        lt = lf + position_embedding_sine(lf.shape[2], lf.shape[3], nf)
        it = it.flatten(2).permute(2, 0, 1) + self.type_embedding[:, :, :, 0]
        lt = lt.flatten(2).permute(2, 0, 1) + self.type_embedding[:, :, :, 1]
        li = it.shape[0]
        tok = self.transformer_encoder(torch.cat([it, lt], 0))
        io = tok[:li].permute(1, 2, 0).reshape(x.shape[0], -1, x.shape[2], x.shape[3])
        lo = tok[li:].permute(1, 2, 0).reshape(lf.shape[0], -1, lf.shape[2], lf.shape[3])
        feats = [self.image_feature_conv(io), self.lidar_feature_conv(lo),
                 self.backbone_route(batch['route_map'].flatten(0, 1)),
                 self.speed_enc(batch['speed'].flatten(0, 1) / self.cfg['SPEED_NORM'])]
        return self.features_combine(torch.cat(feats, -1)).view(b, s, -1)

    def forward(self, batch, noise, use_prior):
        b, s = batch['image'].shape[:2]
        emb = self.encode(batch)
        action = torch.cat([batch['throttle_brake'], batch['steering']], -1)
        prior, post = self.rssm(emb, action, noise, use_prior)
        out = {'prior': prior, 'posterior': post, 'embedding': emb}
        state = torch.cat([post['hidden_state'], post['sample']], -1).flatten(0, 1)
        pol = self.policy(state)
        out['throttle_brake'] = pol[:, :1].view(b, s, 1)
        out['steering'] = pol[:, 1:].view(b, s, 1)
        for dec in self.main_decoders() + self.aux_decoders():
            for k, v in dec(state).items():
                out[k] = v.view(b, s, *v.shape[1:])
        return out

    def forward_deployment(self, batch):
        """Mile.forward(batch, deployment=True) (mile.py:404-489): the recorded `batch['action']`, distribution means instead of
        samples (use_sample=False = zero noise, no prior substitution outside training dropout), then remove_past(state_dict, s)
        (network_utils.py:30-38): only the last time step is kept and decoded (s = 1)."""
        b, s = batch['image'].shape[:2]
        emb = self.encode(batch)
        zero = emb.new_zeros(b, s, 2, self.cfg['STATE_DIM'])
        prior, post = self.rssm(emb, batch['action'].float(), zero, [False] * s)
        prior = {k: v[:, s - 1:].contiguous() for k, v in prior.items()}
        post = {k: v[:, s - 1:].contiguous() for k, v in post.items()}
        out = {'prior': prior, 'posterior': post}
        state = torch.cat([post['hidden_state'], post['sample']], -1).flatten(0, 1)
        pol = self.policy(state)
        out['throttle_brake'] = pol[:, :1].view(b, 1, 1)
        out['steering'] = pol[:, 1:].view(b, 1, 1)
        for dec in self.main_decoders() + self.aux_decoders():
            for k, v in dec(state).items():
                out[k] = v.view(b, 1, *v.shape[1:])
        return out

    def main_decoders(self):
        return tuple(getattr(self, n) for n in ('rgb_decoder', 'lidar_re', 'voxel_decoder') if hasattr(self, n))

    def aux_decoders(self):
        return tuple(getattr(self, n) for n in ('bev_decoder', 'lidar_segmentation', 'sem_image_decoder', 'depth_image_decoder')
                     if hasattr(self, n))


def imagine(model, state, future_horizon, noise):
    """Mile.imagine (mile.py:771-850, predict_action=False): roll the prior forward with the recorded actions and decode
    every imagined state.  state: hidden_state (b,H), sample (b,S), throttle_brake / steering (b,fh,1); noise (b,fh,S)."""
    h, z = state['hidden_state'], state['sample']
    b = h.shape[0]
    states = []
    for t in range(future_horizon):
        a = torch.cat([state['throttle_brake'][:, t], state['steering'][:, t]], -1)
        h, z, _, _ = model.rssm.imagine_step(h, z, a, noise[:, t])
        states.append(torch.cat([h, z], -1))
    st = torch.stack(states, 1)
    flat = st.flatten(0, 1)
    pol = model.policy(flat)
    out = {'state': st, 'throttle_brake': pol[:, :1].view(b, future_horizon, 1),
           'steering': pol[:, 1:].view(b, future_horizon, 1)}
    for dec in model.main_decoders() + model.aux_decoders():
        for k, v in dec(flat).items():
            out[k] = v.view(b, future_horizon, *v.shape[1:])
    return out


# --------------------------------------------------------------------------- losses
def _spatial_regression(pred, target, norm, instance_mask=None):
    """losses.py:74-99 (mask = the given instance mask, else target channel 0 != 255; channel-sum; masked mean)."""
    mask = instance_mask.bool() if instance_mask is not None else target[:, :, :1] != 255
    if mask.sum() == 0:
        return pred.new_zeros(())
    loss = (pred - target).abs() if norm == 1 else (pred - target) ** 2
    loss = loss.sum(dim=-3, keepdim=True)
    return loss[mask].mean()


def _kl(prior_mu, prior_sigma, post_mu, post_sigma):
    """losses.py:102-126."""
    qv, pv = post_sigma[:, 1:] ** 2, prior_sigma[:, 1:] ** 2
    qls, pls = torch.log(post_sigma[:, 1:]), torch.log(prior_sigma[:, 1:])
    kl = pls - qls - 0.5 + (qv + (post_mu[:, 1:] - prior_mu[:, 1:]) ** 2) / (2 * pv)
    # Reference quirk (losses.py:120): the "first timestep" term indexes the ALREADY time-shifted
    # log-sigma / variance ([:, 1:][:, :1] = t=1) but the unshifted mean (t=0).  Reproduced as is.
    first = -qls[:, :1] - 0.5 + (qv[:, :1] + post_mu[:, :1] ** 2) / 2
    return torch.cat([first, kl], 1).sum(-1).mean()


def _sem_scal(logits, target):
    """losses.py:191-251 (logits (N,C,X,Y,Z), target uint8 (N,X,Y,Z))."""
    p_all = F.softmax(logits, dim=1)
    mask = target != 255
    loss, count = 0.0, 0.0
    for i in range(p_all.shape[1]):
        p = p_all[:, i][mask]
        ct = (target[mask] == i).float()
        if ct.sum() > 0:
            count += 1.0
            nom = (p * ct).sum()
            lc = 0.0
            if p.sum() > 0:
                prec = nom / p.sum()
                if 0 <= prec <= 1:
                    lc = lc - torch.log(prec).clamp(min=-100)
            rec = nom / ct.sum()
            if 0 <= rec <= 1:
                lc = lc - torch.log(rec).clamp(min=-100)
            if (1 - ct).sum() > 0:
                spec = ((1 - p) * (1 - ct)).sum() / (1 - ct).sum()
                if 0 <= spec <= 1:
                    lc = lc - torch.log(spec).clamp(min=-100)
            loss = loss + lc
    return loss / count


def _geo_scal(logits, target):
    """losses.py:254-287."""
    p = F.softmax(logits, dim=1)
    empty = p[:, 0]
    mask = target != 255
    ne_t = (target != 0)[mask].float()
    ne_p = (1 - empty)[mask]
    e_p = empty[mask]
    inter = (ne_t * ne_p).sum()
    prec = inter / ne_p.sum()
    rec = inter / ne_t.sum()
    spec = ((1 - ne_t) * e_p).sum() / (1 - ne_t).sum()
    return -(torch.log(prec).clamp(min=-100) + torch.log(rec).clamp(min=-100) + torch.log(spec).clamp(min=-100))


def compute_losses(batch, out, cfg) -> Dict[str, torch.Tensor]:
    """trainer.py:251-390, base_1d branch: 21 keys."""
    L = {}
    L['throttle_brake'] = cfg['W_ACTION'] * (out['throttle_brake'] - batch['throttle_brake']).abs().sum(-1, keepdim=True).mean()
    L['steering'] = cfg['W_ACTION'] * (out['steering'] - batch['steering']).abs().sum(-1, keepdim=True).mean()
    if 'prior' in out and 'posterior' in out:   # absent for imagined outputs (trainer.py:261-265)
        pr, po = out['prior'], out['posterior']
        a = cfg['KL_ALPHA']
        kl = a * _kl(pr['mu'], pr['sigma'], po['mu'].detach(), po['sigma'].detach()) + \
            (1 - a) * _kl(pr['mu'].detach(), pr['sigma'].detach(), po['mu'], po['sigma'])
        L['probabilistic'] = cfg['W_PROB'] * kl
    for f in ((1, 2, 4) if cfg.get('RGB_SUPERVISION', True) else ()):
        d = 1 / f
        rgb = _spatial_regression(out[f'rgb_{f}'], batch[f'rgb_label_{f}'], 1)
        if cfg.get('RGB_INSTANCE'):              # LOSSES.RGB_INSTANCE (trainer.py:303-321): + 0.5 x the same L1 over the instance pixels
            rgb = rgb + 0.5 * _spatial_regression(out[f'rgb_{f}'], batch[f'rgb_label_{f}'], 1, batch[f'image_instance_mask_{f}'])
        L[f'rgb_{f}'] = cfg['W_RGB'] * d * rgb
        if cfg.get('SSIM'):                      # LOSSES.SSIM (trainer.py:312-318): 0.6 * (1 - mean SSIM)
            L[f'ssim_{f}'] = cfg['W_RGB'] * d * (1 - ssim_frames(out[f'rgb_{f}'], batch[f'rgb_label_{f}']).mean()) * 0.6
    for f in (1, 2, 4):
        d = 1 / f
        p, t = out[f'lidar_reconstruction_{f}'], batch[f'range_view_label_{f}']
        L[f'lidar_re_{f}'] = _spatial_regression(p[:, :, :3], t[:, :, :3], 2) * d * cfg['W_LIDAR_RE']
        L[f'lidar_depth_{f}'] = _spatial_regression(p[:, :, -1:], t[:, :, -1:], 1) * d * cfg['W_LIDAR_RE']
    for f in (1, 2, 4):
        d = 1 / f
        logits = out[f'voxel_{f}'].flatten(0, 1)
        tgt = batch[f'voxel_label_{f}'].flatten(0, 1)[:, 0]
        # VoxelLoss (losses.py:144-186): VOXEL_SEG.USE_WEIGHTS -> VOXEL_SEG_WEIGHTS (needs the 9-class head), USE_TOP_K -> mean of the
        # k = int(ratio * voxels) hardest voxels of every frame
        vw = torch.tensor(VOXEL_SEG_WEIGHTS, dtype=logits.dtype) if cfg.get('VOXEL_USE_WEIGHTS') else None
        ce = F.cross_entropy(logits, tgt.long(), reduction='none', weight=vw)
        if cfg.get('VOXEL_USE_TOP_K'):
            bb, ss = out[f'voxel_{f}'].shape[:2]
            ce = ce.view(bb, ss, -1)
            ce = ce.topk(int(cfg.get('VOXEL_TOP_K_RATIO', 0.5) * ce.shape[2]), dim=-1)[0]
        L[f'voxel_{f}'] = d * cfg['W_VOXEL'] * ce.mean()
        L[f'sem_scal_{f}'] = d * cfg['W_VOXEL'] * _sem_scal(logits, tgt)
        L[f'geo_scal_{f}'] = d * cfg['W_VOXEL'] * _geo_scal(logits, tgt)
    # config-off heads (trainer.py:266-291,338-365), when the model produced them
    for f in (1, 2, 4):
        d = 1 / f
        if f'bev_segmentation_{f}' in out:       # SEMANTIC_SEG: top-k 0.25, SEMANTIC_SEG_WEIGHTS; INSTANCE_SEG weights 200 / 0.1
            L[f'bev_segmentation_{f}'] = d * 0.1 * _segmentation_loss(out[f'bev_segmentation_{f}'], batch[f'birdview_label_{f}'], True, 0.25,
                                                                      True, SEMANTIC_SEG_WEIGHTS)
            L[f'bev_center_{f}'] = d * 0.1 * 200.0 * _spatial_regression(out[f'bev_instance_center_{f}'], batch[f'center_label_{f}'], 2)
            L[f'bev_offset_{f}'] = 0.1 * 0.1 * _spatial_regression(out[f'bev_instance_offset_{f}'], batch[f'offset_label_{f}'], 1)
        if f'lidar_segmentation_{f}' in out:      # LIDAR_SEG: top-k 0.5, class weights (config.py:255-260)
            L[f'lidar_seg_{f}'] = _segmentation_loss(out[f'lidar_segmentation_{f}'], batch[f'range_view_seg_label_{f}'], True, 0.5, True) * d * 0.1
        if f'semantic_image_{f}' in out:          # SEMANTIC_IMAGE: no top-k, class weights (config.py:263-268)
            L[f'semantic_image_{f}'] = _segmentation_loss(out[f'semantic_image_{f}'], batch[f'semantic_image_label_{f}'], False, 0.5, True) * d * 0.1
        if f'depth_{f}' in out:
            L[f'depth_{f}'] = _spatial_regression(out[f'depth_{f}'], batch[f'depth_label_{f}'], 1) * d * 0.1
    return L


VOXEL_SEG_WEIGHTS = (1.0, 1.0, 1.0, 1.5, 2.0, 3.0, 1.0, 1.0, 1.0)      # constants.py:39 (is_bev=False, trainer.py:137,161)


def _segmentation_loss(prediction, target, use_top_k, top_k_ratio, use_weights, class_weights=VOXEL_SEG_WEIGHTS):
    """SegmentationLoss.forward (losses.py:22-50) without poly-1."""
    b, s, c, h, w = prediction.shape
    weights = torch.tensor(class_weights, dtype=prediction.dtype) if use_weights else None
    loss = F.cross_entropy(prediction.view(b * s, c, h, w), target.reshape(b * s, h, w).long(), reduction='none', weight=weights)
    loss = loss.view(b, s, -1)
    if use_top_k:
        loss = loss.topk(int(top_k_ratio * loss.shape[2]), dim=-1)[0]
    return torch.mean(loss)


def make_optimizer(model: nn.Module, cfg: dict):
    """trainer.py:1022-1060: 1-D params -> no decay; rest decay 0.01; AdamW lr 1e-4."""
    no_decay, decay = [], []
    for _, p in model.named_parameters():
        (no_decay if p.dim() == 1 else decay).append(p)
    opt = torch.optim.AdamW([{'params': no_decay, 'weight_decay': 0.0},
                             {'params': decay, 'weight_decay': cfg['WEIGHT_DECAY']}], lr=cfg['LR'], weight_decay=0.0)
    # trainer.py:1064-1071: OneCycleLR(max_lr=LR, total_steps=STEPS, pct_start=0.2), stepped every step;
    # constructing it already sets lr = max_lr / 25.
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=cfg['LR'], total_steps=cfg.get('STEPS', 100000),
                                                pct_start=cfg.get('PCT_START', 0.2))
    return opt, sched


def training_step(model: MileRef, raw_batch, noise, use_prior):
    """forward + 21 losses + total (trainer.py:213-231,392-402,511-513). Returns (total, losses, out, batch)."""
    batch = preprocess(raw_batch, model.cfg)
    out = model(batch, noise, use_prior)
    losses = compute_losses(batch, out, model.cfg)
    total = sum(losses.values())
    return total, losses, out, batch


def validation_step(model: MileRef, raw_batch, rf, fh, noise, use_prior, n_samples=2):
    """shared_step(mode='val') (trainer.py:232-249) under no_grad, train-mode BatchNorm (trainer.py:404-409): reconstruct the
    first rf frames, then imagine fh steps (n_samples = PREDICTION.N_SAMPLES times) from the last posterior state with the
    recorded actions.  noise: (b, rf + n_samples*fh, 2, S) — [:, t<rf] feed the observe steps, [:, rf + k*fh + t, 0] step t
    of imagined sample k.  Returns (losses_rf, out_rf, [losses_fh per sample], [out_fh per sample])."""
    with torch.no_grad():
        batch = preprocess(raw_batch, model.cfg)
        brf = {k: v[:, :rf] for k, v in batch.items()}
        bfh = {k: v[:, rf:] for k, v in batch.items()}
        out = model(brf, noise[:, :rf], use_prior[:rf])
        losses = compute_losses(brf, out, model.cfg)
        state = {'hidden_state': out['posterior']['hidden_state'][:, -1], 'sample': out['posterior']['sample'][:, -1],
                 'throttle_brake': batch['throttle_brake'][:, rf:], 'steering': batch['steering'][:, rf:]}
        outs_i, losses_i = [], []
        for k in range(n_samples):
            o = imagine(model, state, fh, noise[:, rf + k * fh:rf + (k + 1) * fh, 0])
            outs_i.append(o)
            losses_i.append(compute_losses(bfh, o, model.cfg))
    return losses, out, losses_i, outs_i


# ------------------------------------------------------------------------------------------------------------------
# Evaluation metrics: CPU restatement of muvo/metrics.py as driven by trainer.py:426-490.
# Pinned against the real reference classes by tests/golden/metrics.json (oracle/refimport/make_golden_metrics.py).
# ------------------------------------------------------------------------------------------------------------------
def ssim_frames(prediction, target, window_size=11, sigma=1.5, L=1.0):
    """losses.py:292-339 (SSIMLoss._ssim): Gaussian window, 'valid' depthwise conv, per-frame mean of the SSIM map.
    prediction/target: (b, s, c, h, w) -> (b*s,) float32."""
    b, s, c, h, w = prediction.shape
    x = torch.arange(window_size)
    g = torch.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2))
    g = (g / g.sum()).unsqueeze(1)
    win = g.mm(g.t()).float()[None, None].expand(c, 1, window_size, window_size).contiguous()
    p, t = prediction.reshape(b * s, c, h, w), target.reshape(b * s, c, h, w)
    c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    mu1, mu2 = F.conv2d(t, win, groups=c), F.conv2d(p, win, groups=c)
    s1 = F.conv2d(t * t, win, groups=c) - mu1 * mu1
    s2 = F.conv2d(p * p, win, groups=c) - mu2 * mu2
    s12 = F.conv2d(t * p, win, groups=c) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return m.mean([1, 2, 3])


def psnr_frames(prediction, target, max_pixel_val=1.0):
    """metrics.py:305-309: per-frame MSE over (c, h, w) -> 20 log10(max / sqrt(mse)); (b, s, c, h, w) -> (b, s)."""
    mse = torch.mean((prediction - target) ** 2, dim=(2, 3, 4))
    return 20 * torch.log10(max_pixel_val / torch.sqrt(mse))


def chamfer_frames(prediction, target):
    """metrics.py:243-249 (CDMetric.add_batch, reducer = mean): Euclidean nearest-neighbour distances both ways;
    (n, P, 3) x (n, Q, 3) -> (n,)."""
    d = torch.cdist(prediction.float(), target.float(), 2)
    return (d.min(1)[0].mean(dim=1) + d.min(2)[0].mean(dim=1)) / 2


def ssc_counts(y_pred, y_true, n_classes):
    """metrics.py:77-100,143-214: with nonempty = (y_true != 255): completion tp/fp/fn on occupied-vs-empty and per-class
    tp/fp/fn over the non-ignored voxels.  Returns (completion[3], tps[C], fps[C], fns[C]) int64."""
    m = y_true != 255
    p, t = y_pred[m].long(), y_true[m].long()
    comp = torch.stack([((t > 0) & (p > 0)).sum(), ((t == 0) & (p > 0)).sum(), ((t > 0) & (p == 0)).sum()])
    tps = torch.stack([((t == j) & (p == j)).sum() for j in range(n_classes)])
    fps = torch.stack([((t != j) & (p == j)).sum() for j in range(n_classes)])
    fns = torch.stack([((t == j) & (p != j)).sum() for j in range(n_classes)])
    return comp, tps, fps, fns


class EvalMetrics:
    """The running statistics the reference keeps per dataloader (trainer.py:426-490; metrics.py SSIMMetric :219-235,
    PSNRMetric :295-317, CDMetric :238-258, SSCMetrics :47-141), including its count = 1e-8 start value."""

    def __init__(self, n_classes=2, scale=50.0):
        self.n_classes, self.scale = n_classes, scale
        self.reset()

    def reset(self):
        self.count = 1e-8
        self.ssim_sum = self.psnr_sum = self.cd_sum = 0.0
        self.comp = torch.zeros(3, dtype=torch.int64)
        self.tps, self.fps, self.fns = (torch.zeros(self.n_classes, dtype=torch.int64) for _ in range(3))

    def add_batch(self, rgb_pred, rgb_target, rv_pred, rv_target, cd_index, voxel_logits, voxel_label):
        self.count += 1
        self.ssim_sum += float(ssim_frames(rgb_pred, rgb_target).mean())
        self.psnr_sum += float(psnr_frames(rgb_pred, rgb_target).mean())
        pt = rv_target.permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * self.scale
        pp = rv_pred.permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * self.scale
        self.cd_sum += float(chamfer_frames(pp[:, cd_index, :-1], pt[:, cd_index, :-1]).mean())
        b, s, c, x, y, z = voxel_logits.shape
        comp, tps, fps, fns = ssc_counts(torch.argmax(voxel_logits.reshape(b * s, c, x, y, z), dim=1),
                                         voxel_label.reshape(b * s, x, y, z), self.n_classes)
        self.comp += comp
        self.tps += tps
        self.fps += fps
        self.fns += fns

    def stats(self):
        tp, fp, fn = (float(v) for v in self.comp)
        iou_ssc = self.tps.float() / (self.tps + self.fps + self.fns + 1e-5).float()
        return dict(ssim=self.ssim_sum / self.count, psnr=self.psnr_sum / self.count, cd=self.cd_sum / self.count,
                    precision=tp / (tp + fp) if tp else 0.0, recall=tp / (tp + fn) if tp else 0.0,
                    iou=tp / (tp + fp + fn) if tp else 0.0, iou_ssc=iou_ssc, iou_ssc_mean=float(iou_ssc[1:].mean()),
                    completion=self.comp.tolist(), tps=self.tps.tolist(), fps=self.fps.tolist(), fns=self.fns.tolist())


# ------------------------------------------------------------------------------------------------------------------
# BEV lifting: CPU restatement of FrustumPooling (muvo/models/frustum_pooling.py:67-217) as called from
# Mile.encode (mile.py:506-522).  Pinned against the real module by tests/golden/frustum_pool.* (make_golden_frustum.py).
# ------------------------------------------------------------------------------------------------------------------
def frustum_grid(size, scale, offsetx):
    """gen_dx_bx (frustum_pooling.py:10-21) + bev_params_to_intrinsics (geometry_utils.py:8-19): cell size dx, first cell
    centre bx, cell counts nx of the (forward, left, up) grid and the metric -> BEV pixel map."""
    bounds = [[-size[0] * scale / 2 - offsetx * scale, size[0] * scale / 2 - offsetx * scale, scale],
              [-size[1] * scale / 2, size[1] * scale / 2, scale], [-10.0, 10.0, 20.0]]
    dx = torch.tensor([r[2] for r in bounds], dtype=torch.float32)
    bx = torch.tensor([r[0] + r[2] / 2.0 for r in bounds], dtype=torch.float32)
    nx = [int(round((r[1] - r[0]) / r[2])) for r in bounds]
    bev = torch.tensor([[1 / scale, 0, size[0] / 2 + offsetx], [0, -1 / scale, size[1] / 2], [0, 0, 1]], dtype=torch.float32)
    return dx, bx, nx, bev


def frustum_cells(intrinsics, extrinsics, H, W, size, scale, offsetx, dbound, downsample):
    """Cell of every frustum point (initialize_frustum :92-106, get_geometry :108-128, voxel_pooling :139-158): returns
    (ix, iy, iz, inside) of shape (B, D, H, W).  The float -> long cast truncates toward zero, so coordinates in (-1, 0)
    land in cell 0 like in the reference."""
    dt = intrinsics.dtype
    dx, bx, nx, bev = frustum_grid(size, scale, offsetx)
    ds = torch.arange(dbound[0], dbound[1], dbound[2], dtype=torch.float32).to(dt)
    D = len(ds)
    xs = torch.linspace(0, W * downsample - 1, W, dtype=torch.float).to(dt).view(1, 1, W).expand(D, H, W)
    ys = torch.linspace(0, H * downsample - 1, H, dtype=torch.float).to(dt).view(1, H, 1).expand(D, H, W)
    dd = ds.view(-1, 1, 1).expand(D, H, W)
    pts = torch.stack((xs * dd, ys * dd, dd), -1).unsqueeze(-1)                      # (D, H, W, 3, 1)
    fx, fy, cx, cy = intrinsics[:, 0, 0], intrinsics[:, 1, 1], intrinsics[:, 0, 2], intrinsics[:, 1, 2]
    one, zero = torch.ones_like(fx), torch.zeros_like(fx)
    kinv = torch.stack((torch.stack((1 / fx, zero, -cx / fx), -1), torch.stack((zero, 1 / fy, -cy / fy), -1),
                        torch.stack((zero, zero, one), -1)), -2)
    combine = extrinsics[:, :3, :3].matmul(kinv)
    g = combine.view(-1, 1, 1, 1, 3, 3).matmul(pts.unsqueeze(0)).squeeze(-1) + extrinsics[:, :3, 3].view(-1, 1, 1, 1, 3)
    bev, dx, bx = bev.to(dt), dx.to(dt), bx.to(dt)
    g0 = (g[..., 0] * bev[0, 0] + bev[0, 2]).long()
    g1 = (g[..., 1] * bev[1, 1] + bev[1, 2]).long()
    g2 = ((g[..., 2] - bx[2] + dx[2] / 2.) / dx[2]).long()
    inside = (g0 >= 0) & (g0 < nx[0]) & (g1 >= 0) & (g1 < nx[1]) & (g2 >= 0) & (g2 < nx[2])
    return g0, g1, g2, inside, nx


def frustum_pool(feat, depth, mask, intrinsics, extrinsics, size, scale, offsetx, dbound, downsample):
    """out[b, c * nz + iz, iy, ix] = sum over the lifted points of the cell of depth[b, d, h, w] * feat[b, c, h, w]
    (outer product mile.py:519, voxel_pooling frustum_pooling.py:130-182, its sort + cumsum + difference being a per-cell
    sum).  mask: (B, D, H, W) bool of the points to lift, or an empty tensor for all (mile.py:511-518)."""
    B, C, H, W = feat.shape
    g0, g1, g2, inside, nx = frustum_cells(intrinsics, extrinsics, H, W, size, scale, offsetx, dbound, downsample)
    if mask.numel():
        inside = inside & mask
    out = torch.zeros(B, C, nx[2], nx[1], nx[0], dtype=feat.dtype)
    b_, d_, h_, w_ = torch.nonzero(inside, as_tuple=True)
    vals = depth[b_, d_, h_, w_].unsqueeze(1) * feat[b_, :, h_, w_]                   # (P, C)
    flat = ((b_ * nx[2] + g2[b_, d_, h_, w_]) * nx[1] + g1[b_, d_, h_, w_]) * nx[0] + g0[b_, d_, h_, w_]
    acc = torch.zeros(B * nx[2] * nx[1] * nx[0], C, dtype=feat.dtype).index_add_(0, flat, vals)
    out = acc.view(B, nx[2], nx[1], nx[0], C).permute(0, 4, 1, 2, 3)
    return torch.cat(out.unbind(dim=2), 1).contiguous()


def frustum_depth_map(depth, dbound, downsample):
    """get_depth_map (frustum_pooling.py:211-217): expected depth, bilinear x downsample (align_corners=False)."""
    ds = torch.arange(dbound[0], dbound[1], dbound[2], dtype=torch.float32).view(1, -1, 1, 1)
    return F.interpolate((ds * depth).sum(1, keepdim=True), scale_factor=float(downsample), mode='bilinear', align_corners=False)


# ------------------------------------------------------------------------------------------------------------------
# Input pipeline: CPU restatement of the lidar range projection and the voxel densification the
# dataset does per frame (muvo/data/dataset.py:275-327, muvo/utils/geometry_utils.py:166-213,
# data/data_preprocessing.py:119-122, constants.py:8,180-204).  Pinned by tests/golden/input_pipeline.npz.
# ------------------------------------------------------------------------------------------------------------------
EGO_VEHICLE_DIMENSION = (4.902, 2.128, 1.511)      # constants.py:8


def label_remap():
    """constants.py:180-204 + dataset.py:281-283: CARLA tag -> {0 empty/sky, 1 occupied}; unknown tags -> 1."""
    import numpy as np
    remap = np.full(23, 1, dtype=np.uint8)
    remap[[0, 13]] = 0
    return remap


def range_projection(points_xyz, obj_tag, lidar_position=(1.0, 0.0, 2.0), fov=(-30, 10), H=64, W=1024):
    """Raw sweep -> (range_view_pcd_xyzd (4, H, W) float32, range_view_pcd_seg (H, W) uint8).  Float64 geometry like numpy's
    promotion in the reference; the closest point of a pixel wins (reference: sort by depth descending, then scatter with
    last-write-wins); equal depths: the lowest point index wins (the reference's order among exact ties is unspecified)."""
    import numpy as np
    pts = points_xyz.astype(np.float32).copy()
    pts += np.asarray(lidar_position)                       # convert_coor_lidar: float32 in-place add, then mirror y
    pts[:, 1] *= -1
    sem = label_remap()[obj_tag]
    x, y, z = EGO_VEHICLE_DIMENSION
    lo, hi = np.array([-x / 2, -y / 2, 0]), np.array([x / 2, y / 2, z])
    keep = ~((lo < pts) & (pts < hi)).all(axis=1)
    pts, sem = pts[keep], sem[keep]
    pc = pts * np.array([1, -1, 1]) - np.asarray(lidar_position, dtype=np.float64)
    depth = np.sqrt((pc * pc).sum(axis=1))
    yaw, pitch = np.arctan2(-pc[:, 1], pc[:, 0]), np.arcsin(pc[:, 2] / depth)
    fd, fu = fov[0] / 180.0 * np.pi, fov[1] / 180.0 * np.pi
    pw = np.clip(np.floor(0.5 * (1.0 - yaw / np.pi) * W), 0, W - 1).astype(np.int64)
    ph = np.clip(np.floor((1.0 - (pitch + abs(fd)) / (fu - fd)) * H), 0, H - 1).astype(np.int64)
    order = np.lexsort((-np.arange(len(depth)), depth))[::-1]   # depth descending; among ties the lowest index comes last
    xyzd = np.zeros((4, H, W), dtype=np.float32)
    xyzd[3] = -1
    seg = np.zeros((H, W), dtype=np.uint8)
    xyzd[3][ph[order], pw[order]] = depth[order]
    for a in range(3):
        xyzd[a][ph[order], pw[order]] = pts[order, a]
    seg[ph[order], pw[order]] = sem[order]
    return xyzd, seg


def voxel_grid(voxel_data, size=(192, 192, 64)):
    """dataset.py:316-327: (Q, 4) rows of x, y, z, CARLA tag -> dense uint8 grid; tag 255 -> 0, remap, later rows win."""
    import numpy as np
    sem = voxel_data[:, -1].copy()
    sem[sem == 255] = 0
    sem = label_remap()[sem]
    vox = np.zeros(size, dtype=np.uint8)
    vox[voxel_data[:, 0], voxel_data[:, 1], voxel_data[:, 2]] = sem
    return vox
