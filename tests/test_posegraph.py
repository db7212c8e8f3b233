"""CPU tests of the sequence data contract: SE(3) pose arithmetic, keyframe and edge rules and the
g2o export are host code, checked here against an independent NumPy/SciPy restatement of the reference's rules
(/root/reference/src/utils/pose6DOF.cpp:98-122,185-190; src/icpslam/icpslam.cpp:70-89,143-152)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from icpslam_amd import sequence, synth


def ref_pose(T):
    T = np.asarray(T, np.float32).astype(np.float64)
    return T[:3, 3].copy(), Rotation.from_matrix(T[:3, :3])


def ref_compose(a, b):
    return a[0] + a[1].apply(b[0]), a[1] * b[1]


def ref_inverse(a):
    return -(a[1].inv().apply(a[0])), a[1].inv()


def quat_close(q, rot, tol=1e-9):
    r = rot.as_quat()
    return min(np.abs(q - r).max(), np.abs(q + r).max()) <= tol


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    return built


def test_pose_from_matrix_compose_inverse():
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = synth.pose_matrix(*rng.uniform(-5, 5, 3), *rng.uniform(-np.pi, np.pi, 3))
        B = synth.pose_matrix(*rng.uniform(-5, 5, 3), *rng.uniform(-np.pi, np.pi, 3))
        a, b = sequence.pose_from_matrix(A), sequence.pose_from_matrix(B)
        pa, qa = sequence.pose_tuple(a)
        ra = ref_pose(A)
        np.testing.assert_allclose(pa, ra[0], atol=1e-12)
        assert quat_close(qa, ra[1], 1e-7) and abs(np.linalg.norm(qa) - 1) < 1e-12      # float32 matrix -> 1e-7
        c = sequence.pose_compose(a, b)
        rc = ref_compose(ref_pose(A), ref_pose(B))
        np.testing.assert_allclose(c.pos, rc[0], atol=1e-6)
        assert quat_close(np.array(c.quat), rc[1], 1e-6)
        inv = sequence.pose_inverse(a)
        ri = ref_inverse(ref_pose(A))
        np.testing.assert_allclose(inv.pos, ri[0], atol=1e-6)
        assert quat_close(np.array(inv.quat), ri[1], 1e-6)
        ident = sequence.pose_compose(a, inv)
        np.testing.assert_allclose(ident.pos, 0, atol=1e-9)


def test_chain_keyframes_edges_and_g2o(tmp_path):
    rng = np.random.default_rng(1)
    Ts, accepted = [], []
    for k in range(60):
        T = synth.pose_matrix(rng.uniform(0.05, 0.2), rng.uniform(-0.02, 0.02), 0.0, 0, 0, rng.uniform(-0.05, 0.05))
        Ts.append(T.astype(np.float32))
        accepted.append(k % 7 != 3)                       # some registrations fail the gate
    g = sequence.PoseGraph(keyframe_distance=0.3)
    kfs = [g.push(T, ok) for T, ok in zip(Ts, accepted)]
    # reference bookkeeping
    pose = (np.zeros(3), Rotation.identity())
    poses, kf_poses, kf_scan, last_kf = [], [], [], None
    for k, (T, ok) in enumerate(zip(Ts, accepted)):
        if not ok:
            assert kfs[k] == -1
            continue
        pose = ref_compose(pose, ref_pose(T))
        poses.append(pose)
        if last_kf is None or np.linalg.norm(pose[0] - last_kf[0]) > 0.3:
            kf_poses.append(pose)
            kf_scan.append(k)
            last_kf = pose
            assert kfs[k] == len(kf_poses) - 1
        else:
            assert kfs[k] == -1
    assert g.num_poses == len(poses) and g.num_keyframes == len(kf_poses) > 5
    for i, p in enumerate(poses):
        pos, q = g.pose(i)
        np.testing.assert_allclose(pos, p[0], atol=1e-6)
        assert quat_close(q, p[1], 1e-6)
    for i, p in enumerate(kf_poses):
        pos, q, s = g.keyframe(i)
        assert s == kf_scan[i]
        np.testing.assert_allclose(pos, p[0], atol=1e-6)
    for i in range(1, len(kf_poses)):
        e = ref_compose(ref_inverse(kf_poses[i]), kf_poses[i - 1])       # new^-1 (+) prev (icpslam.cpp:82)
        pos, q = g.edge(i)
        np.testing.assert_allclose(pos, e[0], atol=1e-6)
        assert quat_close(q, e[1], 1e-6)
    # g2o text
    path = tmp_path / "graph.g2o"
    g.write_g2o(path)
    lines = path.read_text().strip().splitlines()
    verts = [l.split() for l in lines if l.startswith("VERTEX_SE3:QUAT")]
    edges = [l.split() for l in lines if l.startswith("EDGE_SE3:QUAT")]
    assert len(verts) == len(kf_poses) and len(edges) == len(kf_poses) - 1
    for i, v in enumerate(verts):
        assert int(v[1]) == i and len(v) == 9
        np.testing.assert_allclose([float(x) for x in v[2:5]], kf_poses[i][0], atol=1e-6)
    for i, e in enumerate(edges, start=1):
        assert (int(e[1]), int(e[2])) == (i, i - 1) and len(e) == 3 + 7 + 21
        info = np.zeros((6, 6))
        info[np.triu_indices(6)] = [float(x) for x in e[10:]]
        np.testing.assert_allclose(np.diag(info), [0.06, 0.06, 10.0, 0.001, 0.001, 2.0])   # config/icpslam.yaml:21
        assert np.count_nonzero(info - np.diag(np.diag(info))) == 0


def test_first_pose_is_a_keyframe_and_rejects_change_nothing():
    g = sequence.PoseGraph()
    assert g.push(np.eye(4), False) == -1 and g.num_poses == 0
    assert g.push(np.eye(4), True) == 0                    # num_keyframes == 0 -> keyframe (icpslam.cpp:143)
    T = synth.pose_matrix(0.09, 0, 0, 0, 0, 0)
    assert g.push(T, True) == -1 and g.push(T, True) == -1
    assert g.push(T, True) == -1                           # 0.27 m from the last keyframe: not yet
    assert g.push(T, True) == 1                            # 0.36 m > KFS_DIST_THRESH
