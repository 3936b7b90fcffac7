"""Golden fixture of the input pipeline: the REAL reference's lidar conversion, ego-vehicle
masking, label remap and range projection (data/data_preprocessing.py:119-122, muvo/data/dataset.py:275-300,
muvo/utils/geometry_utils.py:166-213) and the sparse-voxel densification (dataset.py:316-327) on deterministic raw inputs;
writes tests/golden/input_pipeline.npz.

Usage: python oracle/refimport/make_golden_input.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.lidar_inputs import lidar_case, voxel_case  # noqa: E402


def main():
    G.import_reference()
    from constants import EGO_VEHICLE_DIMENSION, LABEL_MAP
    from data.data_preprocessing import convert_coor_lidar
    from muvo.utils.geometry_utils import PointCloud
    lidar_position, fov = [1.0, 0.0, 2.0], [-30, 10]            # config.py:85-87
    pts, tag = lidar_case()
    points = convert_coor_lidar(pts.copy(), lidar_position)                                   # dataset.py:278
    remap = np.full((max(LABEL_MAP.keys()) + 1), max(LABEL_MAP.values()), dtype=np.uint8)      # :281-283
    remap[list(LABEL_MAP.keys())] = list(LABEL_MAP.values())
    semantics = remap[tag]
    x, y, z = EGO_VEHICLE_DIMENSION                                                           # :286-290
    ego_box = np.array([[-x / 2, -y / 2, 0], [x / 2, y / 2, z]])
    ego_idx = ((ego_box[0] < points) & (points < ego_box[1])).all(axis=1)
    semantics, points = semantics[~ego_idx], points[~ego_idx]
    pcd = PointCloud(64, 1024, *fov, lidar_position)                                          # trainer.py:117-122
    depth, xyz, sem = pcd.do_range_projection(points, semantics)                              # dataset.py:299
    xyzd = np.concatenate([xyz, depth[..., None]], axis=-1).transpose((2, 0, 1))              # :301-302
    vd = voxel_case()
    voxel_points, voxel_semantics = vd[:, :-1], vd[:, -1].copy()                              # :320-327
    voxel_semantics[voxel_semantics == 255] = 0
    voxel_semantics = remap[voxel_semantics]
    voxels = np.zeros([192, 192, 64], dtype=np.uint8)
    voxels[voxel_points[:, 0], voxel_points[:, 1], voxel_points[:, 2]] = voxel_semantics
    print('points kept', len(points), 'of', len(pts), '; filled pixels', int((depth >= 0).sum()), '; occupied voxels', int((voxels > 0).sum()))
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'input_pipeline.npz'), range_view_pcd_xyzd=xyzd.astype(np.float32),
                        range_view_pcd_seg=sem, voxel=voxels, remap=remap, ego=np.asarray(EGO_VEHICLE_DIMENSION))
    print('wrote tests/golden/input_pipeline.npz')


if __name__ == '__main__':
    main()
