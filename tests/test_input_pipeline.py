"""Input pipeline (lidar range projection and voxel densification, muvo/data/dataset.py:275-327,
geometry_utils.py:166-213).  CPU: the oracle restatement against the arrays the REAL reference functions produced
(tests/golden/input_pipeline.npz, oracle/refimport/make_golden_input.py) — bit-exact.  GPU: the HIP kernels against the
same arrays — bit-exact (index and byte work; the float outputs are copies of input coordinates and one float64 norm)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'input_pipeline.npz')


def test_oracle_input_pipeline_matches_reference():
    from muvo_amd.data.lidar_inputs import lidar_case, voxel_case
    from oracle import muvo_ref as R
    g = np.load(GOLD)
    assert np.array_equal(R.label_remap(), g['remap']) and tuple(g['ego']) == R.EGO_VEHICLE_DIMENSION
    pts, tag = lidar_case()
    xyzd, seg = R.range_projection(pts, tag)
    assert np.array_equal(xyzd, g['range_view_pcd_xyzd']) and np.array_equal(seg, g['range_view_pcd_seg'])
    assert np.array_equal(R.voxel_grid(voxel_case()), g['voxel'])


def test_oracle_input_pipeline_edge_cases():
    from oracle import muvo_ref as R
    xyzd, seg = R.range_projection(np.zeros((0, 3), np.float32), np.zeros(0, np.uint8))
    assert (xyzd[:3] == 0).all() and (xyzd[3] == -1).all() and (seg == 0).all()          # empty sweep: the fill values
    # two points on one ray: the nearer one wins whatever the input order
    ray = np.array([[10.0, 0.5, -1.0], [5.0, 0.25, -0.5]], np.float32)
    a, _ = R.range_projection(ray, np.array([1, 7], np.uint8))
    b, _ = R.range_projection(ray[::-1].copy(), np.array([7, 1], np.uint8))
    assert np.array_equal(a, b) and (a[3] >= 0).sum() == 1 and abs(float(a[3].max()) - np.linalg.norm(ray[1])) < 1e-5
    assert R.voxel_grid(np.zeros((0, 4), np.int64)).sum() == 0
    rows = np.array([[1, 2, 3, 7], [1, 2, 3, 13], [4, 5, 6, 255]], np.int64)             # later row wins; 13 (sky) and 255 -> empty
    v = R.voxel_grid(rows)
    assert v[1, 2, 3] == 0 and v[4, 5, 6] == 0 and v.sum() == 0


@pytest.mark.gpu
def test_hip_input_pipeline_matches_reference(dev):
    from muvo_amd import input_pipeline as IP
    from muvo_amd.data.lidar_inputs import lidar_case, voxel_case
    g = np.load(GOLD)
    pts, tag = lidar_case()
    xyzd, seg = IP.range_projection(torch.from_numpy(pts).to(dev), torch.from_numpy(tag).to(dev))
    assert np.array_equal(xyzd.cpu().numpy(), g['range_view_pcd_xyzd'])
    assert np.array_equal(seg.cpu().numpy(), g['range_view_pcd_seg'])
    vox = IP.voxel_grid(torch.from_numpy(voxel_case()).to(dev))
    assert np.array_equal(vox.cpu().numpy(), g['voxel'])
    # exact depth ties: the lowest index wins, in the restatement and in the kernels
    from oracle import muvo_ref as R
    tie = np.array([[8.0, 1.0, -0.5]] * 3, np.float32)
    tt = np.array([13, 7, 0], np.uint8)
    _, s_hip = IP.range_projection(torch.from_numpy(tie).to(dev), torch.from_numpy(tt).to(dev))
    _, s_ref = R.range_projection(tie, tt)
    assert np.array_equal(s_hip.cpu().numpy(), s_ref) and int(s_ref.sum()) == 0


@pytest.mark.gpu
def test_hip_input_pipeline_full_size(dev):
    """A full sweep (60000 points = POINTS.N_PER_SECOND / CARLA_FPS) and a dense voxel list against the oracle, plus the
    order independence of the projection."""
    from muvo_amd import input_pipeline as IP
    from muvo_amd.data.lidar_inputs import lidar_case, voxel_case
    from oracle import muvo_ref as R
    pts, tag = lidar_case(P=60000, key='lidar_full')
    xyzd, seg = IP.range_projection(torch.from_numpy(pts).to(dev), torch.from_numpy(tag).to(dev))
    rx, rs = R.range_projection(pts, tag)
    assert np.array_equal(xyzd.cpu().numpy(), rx) and np.array_equal(seg.cpu().numpy(), rs)
    perm = np.random.RandomState(0).permutation(len(pts))
    keep = np.ones(len(pts), bool)
    keep[-600:] = False                      # drop the exact duplicates: with them the winner depends on the index by design
    p2, t2 = pts[keep], tag[keep]
    a, _ = IP.range_projection(torch.from_numpy(p2).to(dev), torch.from_numpy(t2).to(dev))
    perm = np.random.RandomState(0).permutation(len(p2))
    b, _ = IP.range_projection(torch.from_numpy(p2[perm]).to(dev), torch.from_numpy(t2[perm]).to(dev))
    assert torch.equal(a, b)
    vd = voxel_case(Q=400000, key='voxel_full')
    assert np.array_equal(IP.voxel_grid(torch.from_numpy(vd).to(dev)).cpu().numpy(), R.voxel_grid(vd))
