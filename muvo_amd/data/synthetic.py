"""Synthetic batches with the reference batch schema (muvo/data/dataset.py:231-369 keys/dtypes).

Values are hash-generated (see utils/detinit.py) so the same (seed, shape) gives the same batch on
any host.  batch k of a run uses seed 1234 + k.
"""
import numpy as np
import torch

from muvo_amd.utils import detinit


def make_batch(b: int, s: int, seed: int = 1234, image_hw=(600, 960), route_hw=(64, 64),
               range_hw=(64, 1024), voxel=(192, 192, 64), n_voxel_classes: int = 2, device='cpu'):
    def key(tag):
        return detinit.name_key(f'batch:{seed}:{tag}')

    out = {}
    n = b * s * 3 * image_hw[0] * image_hw[1]
    out['image'] = torch.from_numpy(
        (detinit.hash_u64(key('image'), n) >> np.uint64(56)).astype(np.uint8).reshape(b, s, 3, *image_hw))
    n = b * s * 3 * route_hw[0] * route_hw[1]
    out['route_map'] = torch.from_numpy(
        (detinit.hash_u64(key('route'), n) >> np.uint64(56)).astype(np.uint8).reshape(b, s, 3, *route_hw))
    # range view: xyz in [0, 50), d = |xyz|, ~10% holes with xyz = 0 and d = -1
    npix = b * s * range_hw[0] * range_hw[1]
    xyz = detinit.uniform_01(key('xyz'), npix * 3).reshape(b, s, range_hw[0], range_hw[1], 3) * np.float32(50.0)
    hole = detinit.uniform_01(key('hole'), npix).reshape(b, s, range_hw[0], range_hw[1]) < np.float32(0.1)
    d = np.sqrt((xyz.astype(np.float64) ** 2).sum(-1)).astype(np.float32)
    xyz[hole] = 0
    d[hole] = -1
    rv = np.concatenate([xyz, d[..., None]], axis=-1).transpose(0, 1, 4, 2, 3)
    out['range_view_pcd_xyzd'] = torch.from_numpy(np.ascontiguousarray(rv))
    nv = b * s * voxel[0] * voxel[1] * voxel[2]
    occ = detinit.uniform_01(key('voxel'), nv) < np.float32(0.1)
    if n_voxel_classes > 2:
        cls = (detinit.hash_u64(key('voxcls'), nv) >> np.uint64(40)) % np.uint64(n_voxel_classes - 1) + np.uint64(1)
        vox = np.where(occ, cls.astype(np.uint8), np.uint8(0))
    else:
        vox = occ.astype(np.uint8)
    out['voxel'] = torch.from_numpy(vox.reshape(b, s, 1, *voxel))
    out['speed'] = torch.from_numpy(detinit.uniform_01(key('speed'), b * s).reshape(b, s, 1) * np.float32(10.0))
    out['throttle_brake'] = torch.from_numpy(detinit.uniform_pm1(key('tb'), b * s).reshape(b, s, 1).copy())
    out['steering'] = torch.from_numpy(detinit.uniform_pm1(key('steer'), b * s).reshape(b, s, 1).copy())
    intr = np.array([[402.8, 0, 480.0], [0, 402.8, 300.0], [0, 0, 1]], dtype=np.float32)
    out['intrinsics'] = torch.from_numpy(np.broadcast_to(intr, (b, s, 3, 3)).copy())
    out['extrinsics'] = torch.from_numpy(np.broadcast_to(np.eye(4, dtype=np.float32), (b, s, 4, 4)).copy())
    if device != 'cpu':
        out = {k: v.to(device) for k, v in out.items()}
    return out


def make_aug_batch(b: int, s: int, seed: int, device='cpu'):
    """make_batch with image-like structure (integer triangle waves + 1/4 hash noise, integer arithmetic only so every host
    builds the same bytes): blur / sharpen / hue act on edges and gradients instead of white noise.  Input of the
    augmentation fixture (tests/golden/augment.*)."""
    batch = make_batch(b, s, seed=seed)

    def tri(t, p):
        return np.abs((t % (2 * p)) - p) * 255 // p

    img = batch['image'].numpy().astype(np.int64)
    x, y = np.arange(img.shape[-1], dtype=np.int64), np.arange(img.shape[-2], dtype=np.int64)
    for f in range(b * s):
        for c in range(3):
            base = (tri(x * (c + 2) + 13 * f, 97)[None, :] * tri(y * (c + 1) + 7 * f, 61)[:, None]) // 255
            img[f // s, f % s, c] = (3 * base + img[f // s, f % s, c]) // 4
    batch['image'] = torch.from_numpy(img.astype(np.uint8))
    r = batch['route_map'].numpy().astype(np.int64)
    h, w = r.shape[-2:]
    ramp = (np.arange(h, dtype=np.int64)[:, None] + np.arange(w, dtype=np.int64)[None, :]) * 255 // (h + w - 2)
    batch['route_map'] = torch.from_numpy(((r + 3 * ramp) // 4).astype(np.uint8))
    if device != 'cpu':
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


def make_noise(b: int, s: int, state_dim: int = 512, seed: int = 1234, use_prior_prob: float = 0.15):
    """RSSM noise eps (b, s, 2, state_dim): [:, :, 0] prior, [:, :, 1] posterior sample noise
    (reference draws: transition.py:179), and the per-timestep 'use prior sample' coin (transition.py:118)."""
    k = detinit.name_key(f'noise:{seed}')
    eps = torch.from_numpy(detinit.normal(k, b * s * 2 * state_dim).reshape(b, s, 2, state_dim).copy())
    coin = detinit.uniform_01(k + 7, s)
    use_prior = [bool(t > 0 and coin[t] < use_prior_prob) for t in range(s)]
    return eps, use_prior


def make_aux_inputs(b, s, seed, image_hw=(600, 960), range_hw=(64, 1024), n_classes=9, device='cpu'):
    """Inputs of the config-off heads: `range_view_pcd_seg` and `semantic_image` class maps (int64,
    values in [0, n_classes)) and a `depth` image in [0, 1) (dataset.py:300-304,331-352)."""
    k = detinit.name_key(f'aux:{seed}')
    n = b * s
    out = {
        'range_view_pcd_seg': torch.from_numpy((detinit.hash_u64(k + 1, n * range_hw[0] * range_hw[1]) % np.uint64(n_classes))
                                               .astype(np.int64)).view(b, s, 1, *range_hw),
        'semantic_image': torch.from_numpy((detinit.hash_u64(k + 2, n * image_hw[0] * image_hw[1]) % np.uint64(n_classes))
                                           .astype(np.int64)).view(b, s, 1, *image_hw),
        'depth': torch.from_numpy(detinit.uniform_01(k + 3, n * image_hw[0] * image_hw[1]).astype(np.float32)).view(b, s, 1, *image_hw),
    }
    return {kk: v.to(device) for kk, v in out.items()} if device != 'cpu' else out


def make_bev_labels(b, s, seed, size=(192, 192), n_classes=8, n_instances=6, device='cpu'):
    """`birdview_label` class map and `instance_label` id map (b, s, 1, H, W) int64 of the SEMANTIC_SEG head (dataset.py:253-273):
    a few axis-aligned boxes per frame carry instance ids 1..n_instances (some absent in some frames), background 0."""
    k = detinit.name_key(f'bev:{seed}')
    n = b * s
    bev = torch.from_numpy((detinit.hash_u64(k + 1, n * size[0] * size[1]) % np.uint64(n_classes)).astype(np.int64)).view(b, s, 1, *size)
    inst = torch.zeros(b, s, 1, *size, dtype=torch.int64)
    r = detinit.hash_u64(k + 2, n * n_instances * 5).reshape(n, n_instances, 5)
    for f in range(n):
        for i in range(n_instances):
            if int(r[f, i, 4] % np.uint64(5)) == 0:
                continue                                          # this instance is not in this frame
            y0, x0 = int(r[f, i, 0] % np.uint64(size[0] - 24)), int(r[f, i, 1] % np.uint64(size[1] - 24))
            hh, ww = 5 + int(r[f, i, 2] % np.uint64(16)), 5 + int(r[f, i, 3] % np.uint64(16))
            inst[f // s, f % s, 0, y0:y0 + hh, x0:x0 + ww] = i + 1
    out = {'birdview_label': bev, 'instance_label': inst}
    return {kk: v.to(device) for kk, v in out.items()} if device != 'cpu' else out


def make_image_instance_mask(b, s, seed, image_hw=(600, 960), n_boxes=5, device='cpu'):
    """`image_instance_mask` (b, s, 1, H, W) bool of LOSSES.RGB_INSTANCE (dataset.py:335-338: vehicle | pedestrian pixels of the
    camera's semantic image): a few axis-aligned boxes per frame."""
    k = detinit.name_key(f'imask:{seed}')
    n = b * s
    r = detinit.hash_u64(k + 1, n * n_boxes * 4).reshape(n, n_boxes, 4)
    m = torch.zeros(b, s, 1, *image_hw, dtype=torch.bool)
    for f in range(n):
        for i in range(n_boxes):
            y0, x0 = int(r[f, i, 0] % np.uint64(image_hw[0] - 120)), int(r[f, i, 1] % np.uint64(image_hw[1] - 160))
            hh, ww = 20 + int(r[f, i, 2] % np.uint64(100)), 30 + int(r[f, i, 3] % np.uint64(130))
            m[f // s, f % s, 0, y0:y0 + hh, x0:x0 + ww] = True
    return m.to(device) if device != 'cpu' else m
