"""CPU tests of the map oracle (oracle/map_oracle.c): the semantics the GPU map is held to."""
import numpy as np

import oracle
from icpslam_amd import synth


def _brute_map(points, res):
    """Independent restatement in NumPy: first point per voxel of the lattice anchored at (first point - res / 2)."""
    pts = np.asarray(points, np.float32)
    fin = np.isfinite(pts[:, :3]).all(1)
    if not fin.any():
        return np.zeros((0, 4), np.float32)
    origin = pts[fin][0, :3].astype(np.float64) - res / 2.0
    out, seen = [], set()
    for p in pts[fin]:
        k = tuple(np.floor((p[:3].astype(np.float64) - origin) / res).astype(np.int64))
        if k not in seen:
            seen.add(k)
            out.append(p)
    return np.array(out, np.float32)


def test_first_point_per_voxel_in_input_order():
    a, b, _ = synth.make_pair(4000, 4000, seed=3)
    m = oracle.VoxelMap(0.5)
    n1 = m.add_points(a)
    n2 = m.add_points(b)
    ref = _brute_map(np.vstack([a, b]), 0.5)
    assert n1 + n2 == len(m) == ref.shape[0]
    assert np.array_equal(m.points().view(np.uint32), ref.view(np.uint32))
    assert m.add_points(a) == 0 and m.add_points(b) == 0          # every voxel is taken now


def test_lattice_is_anchored_at_first_point_minus_half_a_voxel():
    m = oracle.VoxelMap(1.0)
    # anchor (10, 10, 10) -> voxel [9.5, 10.5)^3; 10.4 shares it, 10.6 does not, 9.4 is the voxel below
    pts = np.array([[10, 10, 10, 1], [10.4, 10.4, 10.4, 1], [10.6, 10, 10, 1], [9.4, 10, 10, 1], [9.6, 10.2, 9.9, 1]], np.float32)
    assert m.add_points(pts) == 3
    assert np.array_equal(m.points(), pts[[0, 2, 3]])


def test_pose_is_applied_with_the_transform_contract_and_nonfinite_points_are_skipped():
    a, _, _ = synth.make_pair(3000, 10, seed=4)
    a = a.copy()
    a[0, :3] = np.nan                                              # the anchor is the first FINITE point
    a[5, :3] = np.inf
    T = synth.pose_matrix(1.0, -2.0, 0.3, 0.01, -0.02, 0.4)
    m = oracle.VoxelMap(0.5)
    m.add_points(a, T)
    moved = oracle.transform_cloud(a, T)
    ref = _brute_map(moved, 0.5)
    assert np.array_equal(m.points().view(np.uint32), ref.view(np.uint32))


def test_nn_cloud_is_exact_and_ordered():
    a, b, _ = synth.make_pair(3000, 3000, seed=5)
    T = synth.pose_matrix(0.5, 0.2, 0.0, 0.0, 0.0, 0.1)
    Tinv = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
    m = oracle.VoxelMap(0.5)
    m.add_points(a, T)
    b = b.copy()
    b[7, :3] = np.nan                                              # dropped from the nn cloud
    nn = m.nn_cloud(b, T, Tinv)
    assert nn.shape[0] == b.shape[0] - 1
    mp = m.points()
    q = oracle.transform_cloud(b, T)
    keep = np.isfinite(q[:, :3]).all(1)
    d = ((q[keep, None, :3].astype(np.float64) - mp[None, :, :3]) ** 2).sum(-1)
    want = oracle.transform_cloud(mp[d.argmin(1)], Tinv)
    assert np.allclose(nn, want, atol=1e-6)


def test_empty_map_gives_empty_nn_cloud():
    m = oracle.VoxelMap(0.5)
    a, _, _ = synth.make_pair(100, 10, seed=6)
    assert m.nn_cloud(a, np.eye(4), np.eye(4)).shape[0] == 0
