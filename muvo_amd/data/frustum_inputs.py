"""Deterministic inputs of the BEV lifting operator for fixtures and tests (closed-form hashes, no RNG state)."""
import math

import numpy as np
import torch

from ..utils import detinit


def camera_pose(b, s):
    """(b, s, 4, 4) camera-to-ego poses: camera axes (x right, y down, z forward) -> ego (x forward, y left, z up), mounted
    1.3 m ahead of the ego origin and 1.5 m above ground, with a small frame-dependent yaw."""
    ext = torch.zeros(b, s, 4, 4, dtype=torch.float32)
    r_cam = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    for i in range(b):
        for t in range(s):
            yaw = 0.03 * (t - (s - 1) / 2.0) + 0.01 * i
            r_yaw = torch.tensor([[math.cos(yaw), -math.sin(yaw), 0.0], [math.sin(yaw), math.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
            ext[i, t, :3, :3] = r_yaw @ r_cam
            ext[i, t, :3, 3] = torch.tensor([1.3, 0.0, 1.5])
            ext[i, t, 3, 3] = 1.0
    return ext


def frustum_case(B=2, C=12, H=10, W=26, sparse_count=10, size=(48, 48), scale=0.8, offsetx=-16.0,
                 dbound=(1.0, 38.0, 1.0), downsample=8, key='frustum_case'):
    """One FrustumPooling call as Mile.encode issues it (mile.py:506-522): feature map (B, C, H, W), depth distribution
    (B, D, H, W) = softmax over D, top-k sparse mask, a pinhole camera looking forward (x forward, y left, z up in the ego
    frame) 1.5 m above ground with a small per-frame yaw, and the upstream gradient of the BEV feature map."""
    k = detinit.name_key(key)
    D = len(np.arange(*dbound))
    feat = torch.from_numpy(detinit.uniform_pm1(k + 1, B * C * H * W).astype(np.float32)).view(B, C, H, W)
    logits = torch.from_numpy(detinit.uniform_pm1(k + 2, B * D * H * W).astype(np.float32)).view(B, D, H, W) * 3.0
    depth = logits.softmax(dim=1)
    topk = depth.topk(sparse_count, dim=1)[1]
    mask = torch.zeros(depth.shape, dtype=torch.bool)
    mask.scatter_(1, topk, 1)
    # mile.py:522 passes the mask for the (B, N=1, D, H, W) points
    fw, fh = W * downsample, H * downsample
    f = fw / (2.0 * math.tan(math.radians(100.0) / 2.0))
    intr = torch.tensor([[f, 0.0, fw / 2.0], [0.0, f, fh / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32).repeat(B, 1, 1)
    ext = torch.zeros(B, 4, 4, dtype=torch.float32)
    for b in range(B):
        yaw = 0.05 * (b - (B - 1) / 2.0)
        # camera axes (x right, y down, z forward) -> ego (x forward, y left, z up), then yaw about z
        r_cam = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
        r_yaw = torch.tensor([[math.cos(yaw), -math.sin(yaw), 0.0], [math.sin(yaw), math.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
        ext[b, :3, :3] = r_yaw @ r_cam
        ext[b, :3, 3] = torch.tensor([1.3, 0.1 * b, 1.5])
        ext[b, 3, 3] = 1.0
    nx = (int(round(size[0])), int(round(size[1])))
    gout = torch.from_numpy(detinit.uniform_pm1(k + 3, B * C * nx[1] * nx[0]).astype(np.float32)).view(B, C, nx[1], nx[0])
    return dict(feat=feat, depth=depth, mask=mask, intrinsics=intr, extrinsics=ext, gout=gout, size=size, scale=scale,
                offsetx=offsetx, dbound=list(dbound), downsample=downsample)
