"""GPU parity of the mapper's map (octree_mapper.cpp:55-90,133-172) against oracle/map_oracle.c:
map contents and order bit for bit, nn cloud bit for bit, and the refine-and-grow loop end to end."""
import numpy as np
import pytest

import oracle
from icpslam_amd import NN_BRUTE, NN_GRID, synth
from icpslam_amd.mapper import OctreeMapper, identity_pose
from icpslam_amd.sequence import pose_compose, pose_from_matrix, pose_inverse, pose_to_matrix

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_map_insert_bit_exact_over_several_batches(ctx):
    ref = oracle.VoxelMap(0.5)
    ctx.map_reset(0.5)
    rng = np.random.default_rng(0)
    for k in range(4):
        a, _, _ = synth.make_pair(30000 + 5000 * k, 10, seed=10 + k)
        T = synth.pose_matrix(0.8 * k, 0.1 * k, 0.0, 0.0, 0.0, 0.05 * k) if k else None
        a = a.copy()
        a[rng.integers(0, a.shape[0], 5), :3] = np.nan           # never inserted
        got = ctx.map_add_points(a, T)
        want = ref.add_points(a, T)
        assert got == want and ctx.map_size() == len(ref)
    assert np.array_equal(_bits(ctx.map_points()), _bits(ref.points()))
    # every voxel of the last batch is taken now
    assert ctx.map_add_points(a, T) == 0


def test_map_anchor_is_first_finite_point_and_other_resolutions(ctx):
    a, _, _ = synth.make_pair(20000, 10, seed=20)
    a = a.copy()
    a[:3, :3] = np.nan
    for res in (0.05, 0.2, 3.0):
        ref = oracle.VoxelMap(res)
        ctx.map_reset(res)
        assert ctx.map_add_points(a) == ref.add_points(a)
        assert np.array_equal(_bits(ctx.map_points()), _bits(ref.points()))


def test_map_duplicates_and_single_voxel(ctx):
    pts = np.ones((5000, 4), np.float32)
    pts[:, :3] = (1.0, 2.0, 3.0)
    ctx.map_reset(0.5)
    assert ctx.map_add_points(pts) == 1
    assert np.array_equal(ctx.map_points(), pts[:1])
    assert ctx.map_add_points(np.zeros((0, 4), np.float32)) == 0


@pytest.mark.parametrize("mode", [NN_GRID, NN_BRUTE])
def test_nn_target_bit_exact(ctx, mode):
    scan0, scan1, Tgt = synth.make_pair(40000, 40000, seed=30)
    pose = synth.pose_matrix(2.0, -1.0, 0.1, 0.01, 0.02, 0.3)
    pose_inv = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    ref = oracle.VoxelMap(0.5)
    ref.add_points(scan1, pose)
    ctx.set_params(ctx.default_params(), nn_mode=mode)
    ctx.map_reset(0.5)
    ctx.map_add_points(scan1, pose)
    src = scan0.copy()
    src[11, :3] = np.nan                                           # dropped
    src[12, :3] = (500.0, 500.0, 50.0)                             # far from the map: brute-force completion
    ctx.set_source(src)
    nn = ctx.map_nn_target(pose, pose_inv)
    want = ref.nn_cloud(src, pose, pose_inv)
    assert nn.shape == want.shape == (src.shape[0] - 1, 4)
    assert np.array_equal(_bits(nn), _bits(want))
    # the nn cloud IS the target now: aligning against it equals aligning against the oracle's nn cloud
    r = ctx.align()
    o = oracle.icp_align(src, want, oracle.default_params())
    assert r["iterations"] == o["iterations"] and r["n_corr"] == o["n_corr"]
    assert np.abs(r["T"][:3, :3] - o["T"][:3, :3]).max() <= 1e-4   # BASELINE tolerance (R)
    assert np.linalg.norm(r["T"][:3, 3] - o["T"][:3, 3]) <= 1e-3   # BASELINE tolerance (t), metres


def test_nn_target_on_empty_map(ctx):
    a, _, _ = synth.make_pair(1000, 10, seed=31)
    ctx.map_reset(0.5)
    ctx.set_source(a)
    nn = ctx.map_nn_target(np.eye(4), np.eye(4))
    assert nn.shape == (0, 4)
    r = ctx.align()                                                 # PCL: empty target -> not converged, identity
    assert not r["converged"] and np.array_equal(r["T"], np.eye(4, dtype=np.float32))


def test_refine_and_grow_matches_the_oracle_flow(ctx):
    """octree_mapper.cpp:133-172 over five scans of a short drive, odometry = ground truth + a known error."""
    scene = synth.make_scene(seed=40)
    poses = [synth.pose_matrix(0.6 * k, 0.05 * k, 0.0, 0.0, 0.0, 0.02 * k) for k in range(5)]
    scans = [synth.scan(scene, P, 30000, seed=400 + k) for k, P in enumerate(poses)]
    err = synth.pose_matrix(0.15, -0.1, 0.02, 0.0, 0.0, 0.01)

    from icpslam_amd import P2P_SVD
    mapper = OctreeMapper(ctx, octree_resolution=0.5, method=P2P_SVD)      # (the default is GICP, as octree_mapper.cpp:104 has it)
    ref = oracle.VoxelMap(0.5)
    p_icp = oracle.default_params(max_iterations=30)
    for k, (scan, P) in enumerate(zip(scans, poses)):
        raw = pose_from_matrix((P.astype(np.float64) @ err.astype(np.float64)).astype(np.float32) if k else P)
        raw_M = pose_to_matrix(raw)
        raw_Minv = pose_to_matrix(pose_inverse(raw))
        ok, transform, refined, info = mapper.refineTransformAndGrowMap(scan, raw)
        # the same sequence on the CPU
        if len(ref) == 0:
            ref.add_points(scan, raw_M)
            assert not ok and info["seeded"]
        else:
            nn = ref.nn_cloud(scan, raw_M, raw_Minv)
            o = oracle.icp_align(scan, nn, p_icp)
            assert ok == o["converged"]
            assert info["n_nn"] == nn.shape[0]
            T_gpu = pose_to_matrix(transform)
            assert np.abs(T_gpu[:3, :3] - o["T"][:3, :3]).max() <= 1e-4
            assert np.linalg.norm(T_gpu[:3, 3] - o["T"][:3, 3]) <= 1e-3
            # the CPU map grows with the GPU's refined pose so that both maps stay comparable bit for bit
            ref.add_points(scan, pose_to_matrix(refined))
            # the refinement must undo most of the injected odometry error
            resid = np.linalg.inv(P.astype(np.float64)) @ pose_to_matrix(refined).astype(np.float64)
            assert np.linalg.norm(resid[:3, 3]) < 0.08
        assert mapper.map_size == len(ref)
    assert np.array_equal(_bits(mapper.map_cloud()), _bits(ref.points()))


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_map_against_oracle(ctx, seed):
    """Random resolutions, poses, batch sizes and shapes (clumps, sheets, duplicates, non-finite points): the map and the
    nn cloud stay bit-identical to the sequential oracle."""
    rng = np.random.default_rng(500 + seed)
    res = float(rng.choice([0.07, 0.25, 0.5, 1.3, 4.0]))
    ref = oracle.VoxelMap(res)
    ctx.set_params(ctx.default_params())
    ctx.map_reset(res)
    for k in range(int(rng.integers(2, 5))):
        n = int(rng.integers(50, 40000))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            p = rng.uniform(-25, 25, (n, 3))
        elif kind == 1:
            p = rng.uniform(-25, 25, (n, 3))
            p[:, 2] = rng.normal(0.0, 0.02, n)                    # a sheet: many points per voxel column
        else:
            c = rng.uniform(-10, 10, (8, 3))
            p = c[rng.integers(0, 8, n)] + rng.normal(0, 0.3, (n, 3))
        cloud = np.ones((n, 4), np.float32)
        cloud[:, :3] = p.astype(np.float32)
        cloud[rng.integers(0, n, 3), :3] = np.nan
        cloud[n // 2: n // 2 + 20] = cloud[n // 2]                 # exact duplicates
        pose = synth.pose_matrix(*rng.uniform(-3, 3, 3), *rng.uniform(-0.3, 0.3, 3))
        assert ctx.map_add_points(cloud, pose) == ref.add_points(cloud, pose)
    assert ctx.map_size() == len(ref)
    assert np.array_equal(_bits(ctx.map_points()), _bits(ref.points()))
    q = np.ones((3000, 4), np.float32)
    q[:, :3] = rng.uniform(-30, 30, (3000, 3)).astype(np.float32)
    q[7, :3] = np.inf
    pose = synth.pose_matrix(*rng.uniform(-1, 1, 3), *rng.uniform(-0.2, 0.2, 3))
    pinv = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    ctx.set_source(q)
    assert np.array_equal(_bits(ctx.map_nn_target(pose, pinv)), _bits(ref.nn_cloud(q, pose, pinv)))


@pytest.mark.parametrize("seed,n_scans,res", [(7, 3, 0.5), (11, 4, 0.5), (13, 3, 0.3), (17, 2, 1.0)])
def test_pcl_approx_search_mode_returns_the_restatements_nn_cloud(built, seed, n_scans, res):
    """icpgpu_map_set_search(PCL_APPROX): the nn cloud of OctreePointCloudSearch::approxNearestSearch as the reference
    literally calls it (octree_mapper.cpp:84) -- the octree's bounding box grown point by point, the greedy descent by voxel
    centre -- equals oracle/map_approx_np.py's (the NumPy restatement of PCL 1.8's octree) bit for bit, map after map as the
    box doubles; switching back gives the exact neighbours again."""
    from icpslam_amd import Context
    from oracle.map_approx_np import ApproxOctreeMap
    rng = np.random.default_rng(seed)
    scene = synth.make_scene(70 + seed)
    poses = [np.eye(4)]
    for _ in range(n_scans):
        poses.append(poses[-1] @ synth.pose_matrix(rng.uniform(-0.6, 0.6), rng.uniform(-0.3, 0.3), 0.0, 0.0, 0.0,
                                                   np.deg2rad(rng.uniform(-4, 4))))
    scans = [synth.scan(scene, P, 9000, seed=900 + 10 * seed + k) for k, P in enumerate(poses)]
    scans[1][5, :3] = np.nan                                   # a non-finite query / map candidate along the way
    am = ApproxOctreeMap(res)
    with Context(0) as c:
        c.map_reset(res)
        c.map_set_search(True)
        for k in range(n_scans):
            P = poses[k].astype(np.float32)
            added = c.map_add_points(scans[k], P)
            assert added == am.add_points(oracle.transform_cloud(scans[k], P))
            assert np.array_equal(c.map_points(), am.map_points())
            raw = poses[k + 1].copy()
            raw[:3, 3] += (0.07, -0.04, 0.01)
            raw = raw.astype(np.float32)
            raw_inv = np.linalg.inv(raw.astype(np.float64)).astype(np.float32)
            q = oracle.transform_cloud(scans[k + 1], raw)
            fin = np.isfinite(q[:, :3]).all(axis=1)
            ia = am.nn_indices_approx(q[fin])
            want = oracle.transform_cloud(am.map_points()[ia], raw_inv)
            c.set_source(scans[k + 1])
            got = c.map_nn_target(raw, raw_inv)
            assert got.shape == want.shape and np.array_equal(got, want), (k, am.depth)
        assert am.depth >= 6                                   # the box has doubled a few times on the way
        c.map_set_search(False)                                # ... and the exact search still is what it was
        ie, _ = oracle.nn(scans[n_scans], am.map_points(), raw)
        got = c.map_nn_target(raw, raw_inv)
        assert np.array_equal(got, oracle.transform_cloud(am.map_points()[ie[fin]], raw_inv))
