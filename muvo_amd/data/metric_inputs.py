"""Deterministic synthetic inputs for the evaluation metrics (shared by the golden-fixture generator, the oracle tests and
the GPU tests): closed-form hashes, no RNG state."""
import numpy as np
import torch

from ..utils import detinit


def metric_case(k: int, b=1, s=2, h=40, w=56, rh=8, rw=64, vox=(12, 10, 8), n_pts=96):
    """Batch k of a small evaluation run: rgb prediction/target in [0, 1], range-view xyz+depth prediction/target (already
    divided by LIDAR_RE.SCALE as the model outputs them), voxel logits + labels with a few 255 (ignore) voxels, and the
    point subset the Chamfer metric draws (the reference draws it with np.random.randint, trainer.py:455)."""
    key = detinit.name_key(f'metric_case:{k}')
    n = b * s
    tgt = torch.from_numpy(detinit.uniform_01(key + 1, n * 3 * h * w).astype(np.float32)).view(b, s, 3, h, w)
    noise = torch.from_numpy(detinit.uniform_pm1(key + 2, n * 3 * h * w).astype(np.float32)).view(b, s, 3, h, w)
    pred = (tgt + 0.15 * noise).clamp(0.0, 1.0)
    rv_t = torch.from_numpy(detinit.uniform_pm1(key + 3, n * 4 * rh * rw).astype(np.float32)).view(b, s, 4, rh, rw)
    rv_p = rv_t + 0.05 * torch.from_numpy(detinit.uniform_pm1(key + 4, n * 4 * rh * rw).astype(np.float32)).view(b, s, 4, rh, rw)
    idx = (detinit.hash_u64(key + 5, n_pts) % np.uint64(rh * rw)).astype(np.int64)
    x, y, z = vox
    logits = torch.from_numpy(detinit.uniform_pm1(key + 6, n * 2 * x * y * z).astype(np.float32)).view(b, s, 2, x, y, z)
    u = detinit.uniform_01(key + 7, n * x * y * z)
    lab = (u < 0.3).astype(np.uint8)
    lab[u > 0.97] = 255
    label = torch.from_numpy(lab).view(b, s, x, y, z)
    return dict(rgb_pred=pred, rgb_target=tgt, rv_pred=rv_p, rv_target=rv_t, cd_index=idx, voxel_logits=logits, voxel_label=label)
