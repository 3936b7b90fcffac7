"""Input pipeline on the GPU: the per-frame lidar and voxel preparation of the reference's dataset
(muvo/data/dataset.py:275-327, muvo/utils/geometry_utils.py:166-213) as HIP kernels, so raw sweeps / sparse voxel lists can
be handed to the device instead of being projected on the host (the reference spends ~1.8 s of CPU per frame on input
preparation)."""
import ctypes as C

import numpy as np
import torch

from . import ops

EGO_VEHICLE_DIMENSION = (4.902, 2.128, 1.511)      # constants.py:8


def label_remap(device):
    """constants.py:180-204 + dataset.py:281-283 as a 256-entry table (unknown tags -> 1 = occupied)."""
    t = torch.ones(256, dtype=torch.uint8)
    t[0] = 0
    t[13] = 0
    return t.to(device)


def range_projection(points_xyz, obj_tag, lidar_position=(1.0, 0.0, 2.0), fov=(-30, 10), H=64, W=1024, with_seg=True):
    """points_xyz (P, 3) float32 device tensor in the lidar frame, obj_tag (P,) uint8 -> (range_view_pcd_xyzd (4, H, W) float32,
    range_view_pcd_seg (H, W) uint8 or None)."""
    pts, tag = points_xyz.float().contiguous(), obj_tag.to(torch.uint8).contiguous()
    dev = pts.device
    xyzd = torch.empty(4, H, W, device=dev, dtype=torch.float32)
    seg = torch.empty(H, W, device=dev, dtype=torch.uint8) if with_seg else None
    scratch = torch.empty(H * W * 3, device=dev, dtype=torch.int32)
    lp = (C.c_double * 3)(*lidar_position)
    ego = (C.c_double * 3)(*EGO_VEHICLE_DIMENSION)
    ops._ck(ops.lib().muvo_range_projection(ops._f(pts), ops._p(tag), ops._p(label_remap(dev)), ops._i64(pts.shape[0]), lp, ego,
                                           C.c_double(fov[0]), C.c_double(fov[1]), H, W, ops._p(scratch), ops._f(xyzd), ops._p(seg),
                                           ops._st()))
    return xyzd, seg


def voxel_grid(voxel_data, size=(192, 192, 64)):
    """voxel_data (Q, 4) int64 device tensor of x, y, z, CARLA tag -> dense uint8 grid of `size` (dataset.py:316-327)."""
    rows = voxel_data.to(torch.int64).contiguous()
    dev = rows.device
    n = size[0] * size[1] * size[2]
    scratch = torch.empty(n, device=dev, dtype=torch.int32)
    vox = torch.empty(size, device=dev, dtype=torch.uint8)
    ops._ck(ops.lib().muvo_voxel_grid(ops._p(rows), ops._i64(rows.shape[0]), ops._p(label_remap(dev)), size[0], size[1], size[2],
                                     ops._p(scratch), ops._p(vox), ops._st()))
    return vox
