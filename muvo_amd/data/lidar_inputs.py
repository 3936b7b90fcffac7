"""Deterministic raw inputs of the input pipeline for fixtures and tests: a lidar sweep in CARLA's sensor
frame with object tags, and a sparse voxel list with duplicates and 255 labels."""
import numpy as np

from ..utils import detinit


def lidar_case(P=20000, key='lidar_case'):
    """points_xyz float32 (P, 3) in the lidar frame (before convert_coor_lidar), ObjTag uint8 (P,) in [0, 22]; a share of the
    points falls into the ego-vehicle box, some are exact duplicates of others (same pixel, same depth, same tag)."""
    k = detinit.name_key(key)
    r = 1.5 + 48.0 * detinit.uniform_01(k + 1, P) ** 2
    yaw = np.pi * detinit.uniform_pm1(k + 2, P)
    pitch = np.deg2rad(-30.0 + 40.0 * detinit.uniform_01(k + 3, P))
    pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch)], axis=1).astype(np.float32)
    n_ego = P // 50
    pts[:n_ego] = (detinit.uniform_pm1(k + 4, n_ego * 3).reshape(n_ego, 3) * np.array([2.0, 0.9, 0.7]) + np.array([-1.0, 0.0, -1.3])).astype(np.float32)
    pts[-P // 100:] = pts[P // 2:P // 2 + P // 100]                     # exact duplicates
    tag = (detinit.hash_u64(k + 5, P) % np.uint64(23)).astype(np.uint8)
    # same tag on the duplicates: which of two points at exactly equal depth wins is an accident of numpy's unstable
    # argsort in the reference (the restatement and the kernels define it: the lowest index)
    tag[-P // 100:] = tag[P // 2:P // 2 + P // 100]
    return pts, tag


def voxel_case(Q=30000, size=(192, 192, 64), key='voxel_case'):
    """voxel_data int64 (Q, 4): x, y, z, CARLA semantic tag (0..22 or 255), with repeated coordinates (later rows win)."""
    k = detinit.name_key(key)
    xyz = np.stack([(detinit.hash_u64(k + 1 + a, Q) % np.uint64(size[a])).astype(np.int64) for a in range(3)], axis=1)
    xyz[Q // 2:Q // 2 + Q // 20] = xyz[:Q // 20]                         # duplicates
    sem = (detinit.hash_u64(k + 7, Q) % np.uint64(23)).astype(np.int64)
    sem[detinit.uniform_01(k + 8, Q) > 0.98] = 255
    return np.concatenate([xyz, sem[:, None]], axis=1)
