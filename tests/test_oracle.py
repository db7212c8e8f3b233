"""CPU tests: pin the C oracle (oracle/icp_oracle.c).

The reference has no tests for this path (parity unpinned), so the oracle is pinned by
  (i) construction-known ground truth, (ii) the independent NumPy/SciPy restatement, (iii) the committed golden
  fixtures, (iv) properties (permutation invariance, rigid equivariance, identity, kd-tree == brute force).
"""
import glob
import os

import numpy as np
import pytest

import oracle
from icpslam_amd import synth
from oracle import icp_oracle_np as onp

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def dR(a, b):
    return float(np.abs(np.asarray(a, np.float64)[:3, :3] - np.asarray(b, np.float64)[:3, :3]).max())


def dt(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64)[:3, 3] - np.asarray(b, np.float64)[:3, 3]))


@pytest.fixture(scope="module", autouse=True)
def _build():
    oracle.build()


def test_golden_files_present():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(path):
    g = np.load(path)
    p = oracle.default_params(max_iterations=int(g["max_iterations"]), max_correspondence_distance=float(g["max_corr"]))
    guess = g["guess"] if bool(g["has_guess"]) else None
    r = oracle.icp_align(g["src"], g["tgt"], p, guess=guess, want_fitness=True, want_trace=True)
    assert r["converged"] == bool(g["converged"])
    assert r["iterations"] == int(g["iterations"])
    assert r["state"] == int(g["state"])
    assert r["n_corr"] == int(g["n_corr"])
    assert dR(r["T"], g["T"]) <= 1e-6 and dt(r["T"], g["T"]) <= 1e-6
    assert abs(r["fitness"] - float(g["fitness"])) <= 1e-6 * max(1.0, float(g["fitness"]))
    assert [t["n_corr"] for t in r["trace"]] == list(g["trace_n_corr"])
    for k, t in enumerate(r["trace"]):
        np.testing.assert_allclose(t["Tk"], g["trace_Tk"][k], atol=1e-6)
        np.testing.assert_allclose(t["final"], g["trace_final"][k], atol=1e-6)
        assert abs(t["mse"] - g["trace_mse"][k]) <= 1e-7 * max(1.0, g["trace_mse"][k])
    # bit-level pin of the NN arithmetic contract
    idx, d2 = oracle.nn(g["src"], g["tgt"], np.eye(4), nn_mode=oracle.NN_KDTREE)
    assert np.array_equal(idx, g["nn_idx"]) and np.array_equal(d2.view(np.uint32), g["nn_d2"].view(np.uint32))


@pytest.mark.parametrize("n_s,n_t,seed", [(1, 1, 0), (50, 3, 1), (2000, 2000, 2), (3000, 777, 3), (100, 20000, 4)])
def test_kdtree_equals_bruteforce(n_s, n_t, seed):
    rng = np.random.default_rng(seed)
    src = np.ones((n_s, 4), np.float32)
    tgt = np.ones((n_t, 4), np.float32)
    src[:, :3] = rng.uniform(-30, 30, (n_s, 3))
    tgt[:, :3] = np.round(rng.uniform(-30, 30, (n_t, 3)), 1)     # rounding creates duplicate coordinates / ties
    T = synth.pose_matrix(0.3, -0.2, 0.1, 0.01, -0.02, 0.05)
    for arith in (oracle.ARITH_FMA, oracle.ARITH_FLANN):
        ik, dk = oracle.nn(src, tgt, T, oracle.NN_KDTREE, arith)
        ib, db = oracle.nn(src, tgt, T, oracle.NN_BRUTE, arith)
        assert np.array_equal(ik, ib) and np.array_equal(dk.view(np.uint32), db.view(np.uint32))


def test_nn_against_float64_reference():
    src, tgt, _ = synth.make_pair(3000, 3000, seed=5)
    idx, d2 = oracle.nn(src, tgt)
    d = np.linalg.norm(src[:, None, :3].astype(np.float64) - tgt[None, :, :3].astype(np.float64), axis=2) ** 2
    best = d.min(axis=1)
    np.testing.assert_allclose(d2, best, rtol=1e-5, atol=1e-9)
    chosen = d[np.arange(3000), idx]
    np.testing.assert_allclose(chosen, best, rtol=1e-5, atol=1e-9)


def test_svd3_reconstructs_and_is_orthonormal():
    rng = np.random.default_rng(0)
    mats = [rng.normal(size=(3, 3)) for _ in range(50)]
    mats += [np.outer(rng.normal(size=3), rng.normal(size=3)), np.zeros((3, 3)), np.diag([3.0, 1e-9, 0.0]),
             np.eye(3), -np.eye(3)]
    for A in mats:
        U, s, V = oracle.svd3(A)
        np.testing.assert_allclose(U @ np.diag(s) @ V.T, A, atol=1e-12)
        np.testing.assert_allclose(U.T @ U, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(V.T @ V, np.eye(3), atol=1e-12)
        assert s[0] >= s[1] >= s[2] >= 0
        np.testing.assert_allclose(s, np.linalg.svd(A, compute_uv=False), atol=1e-12)


def test_umeyama_matches_numpy_and_handles_reflection():
    rng = np.random.default_rng(1)
    for trial in range(20):
        p = rng.normal(size=(200, 3)) * (5, 3, 0.5)
        T = synth.pose_matrix(*rng.uniform(-1, 1, 3), *rng.uniform(-0.3, 0.3, 3))
        q = p @ T[:3, :3].T + T[:3, 3] + rng.normal(size=(200, 3)) * 0.01
        if trial % 4 == 0:
            q[:, 2] = 0.0    # planar target: the reflection branch of umeyama may trigger
            p[:, 2] = 0.0
        sums = np.zeros(17)
        sums[0] = 200
        sums[1:4] = p.sum(0)
        sums[4:7] = q.sum(0)
        sums[7:16] = (q.T @ p).reshape(-1)
        Tk = oracle.umeyama(sums)
        ref = onp.umeyama(p, q)
        np.testing.assert_allclose(Tk, ref, atol=1e-9)
        assert abs(np.linalg.det(Tk[:3, :3]) - 1.0) < 1e-9


@pytest.mark.parametrize("seed,n", [(11, 3000), (12, 5000)])
def test_c_oracle_matches_numpy_restatement(seed, n):
    src, tgt, _ = synth.make_pair(n, n, seed=seed)
    for iters in (10, 30):
        a = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=iters), want_fitness=True, want_trace=True)
        b = onp.icp_align(src, tgt, max_iterations=iters, want_fitness=True)
        assert a["iterations"] == b["iterations"] and a["state"] == b["state"] and a["n_corr"] == b["n_corr"]
        assert dR(a["T"], b["T"]) <= 1e-6 and dt(a["T"], b["T"]) <= 1e-6
        assert [t["n_corr"] for t in a["trace"]] == [t["n_corr"] for t in b["trace"]]
        assert abs(a["fitness"] - b["fitness"]) <= 1e-6 * max(1.0, b["fitness"])


def test_c_oracle_matches_numpy_restatement_50k():
    """The same at BASELINE config 2's size (50k x 50k); and the PCL-float / FLANN-ordered flavour of the C oracle stays
    within the BASELINE tolerance of the contract there too."""
    src, tgt, _ = synth.make_pair(50000, 50000, seed=2)
    a = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=10), want_fitness=True, want_trace=True)
    b = onp.icp_align(src, tgt, max_iterations=10, want_fitness=True)
    assert a["iterations"] == b["iterations"] and a["state"] == b["state"] and a["n_corr"] == b["n_corr"]
    assert dR(a["T"], b["T"]) <= 1e-6 and dt(a["T"], b["T"]) <= 1e-6
    assert [t["n_corr"] for t in a["trace"]] == [t["n_corr"] for t in b["trace"]]
    assert abs(a["fitness"] - b["fitness"]) <= 1e-6 * max(1.0, b["fitness"])
    c = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=10, precision=oracle.PREC_PCL_F32, arith=oracle.ARITH_FLANN))
    assert dR(a["T"], c["T"]) <= 1e-4 and dt(a["T"], c["T"]) <= 1e-3


def test_known_answer_recovers_ground_truth():
    src, tgt, T_gt = synth.make_known_answer_pair(5000, seed=21)
    r = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=50))
    assert r["converged"] and r["state"] == 2
    assert dR(r["T"], T_gt) <= 1e-5 and dt(r["T"], T_gt) <= 1e-4
    assert r["n_corr"] == 5000


def test_identity_on_identical_clouds():
    src, _, _ = synth.make_pair(2000, 10, seed=22)
    r = oracle.icp_align(src, src.copy(), want_fitness=True)
    assert r["converged"] and r["iterations"] == 1 and r["state"] == 2
    np.testing.assert_allclose(r["T"], np.eye(4), atol=1e-6)
    assert r["fitness"] == 0.0


def test_permutation_invariance_and_rigid_equivariance():
    src, tgt, _ = synth.make_pair(3000, 3000, seed=23)
    base = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30))
    rng = np.random.default_rng(0)
    perm = oracle.icp_align(src[rng.permutation(3000)], tgt[rng.permutation(3000)], oracle.default_params(max_iterations=30))
    assert perm["iterations"] == base["iterations"] and perm["n_corr"] == base["n_corr"]
    assert dR(perm["T"], base["T"]) <= 1e-6 and dt(perm["T"], base["T"]) <= 1e-6
    # move both clouds by the same rigid motion G: T' = G T G^-1
    G = synth.pose_matrix(3.0, -2.0, 0.5, 0.02, -0.01, 0.7)
    s2, t2 = oracle.transform_cloud(src, G), oracle.transform_cloud(tgt, G)
    moved = oracle.icp_align(s2, t2, oracle.default_params(max_iterations=30))
    expect = G @ base["T"].astype(np.float64) @ np.linalg.inv(G)
    assert moved["iterations"] == base["iterations"]
    assert dR(moved["T"], expect) <= 2e-4 and dt(moved["T"], expect) <= 2e-3   # float32 re-quantisation of the inputs


def test_mse_non_increasing_and_modes_agree():
    src, tgt, _ = synth.make_pair(4000, 4000, seed=24)
    a = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30), want_trace=True)
    mses = [t["mse"] for t in a["trace"]]
    assert all(m2 <= m1 * (1 + 1e-3) + 1e-9 for m1, m2 in zip(mses, mses[1:]))
    # PCL-float flavour and FLANN (no-FMA) distance order stay within the parity tolerance of the f64/FMA contract
    b = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30, precision=oracle.PREC_PCL_F32))
    c = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30, arith=oracle.ARITH_FLANN))
    d = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30, nn_mode=oracle.NN_BRUTE))
    assert dR(a["T"], b["T"]) <= 1e-4 and dt(a["T"], b["T"]) <= 1e-3
    assert dR(a["T"], c["T"]) <= 1e-4 and dt(a["T"], c["T"]) <= 1e-3
    assert np.array_equal(a["T"], d["T"]) and a["n_corr"] == d["n_corr"]


def test_degenerate_inputs():
    src, tgt, _ = synth.make_pair(500, 500, seed=25)
    e = np.zeros((0, 4), np.float32)
    r = oracle.icp_align(src, e)
    assert not r["converged"] and r["iterations"] == 0 and np.array_equal(r["T"], np.eye(4, dtype=np.float32))
    r = oracle.icp_align(e, tgt)
    assert not r["converged"] and r["state"] == 5
    far = src.copy()
    far[:, 0] += 1000
    r = oracle.icp_align(far, tgt, want_fitness=True)
    assert not r["converged"] and r["state"] == 5 and r["n_corr"] == 0 and r["fitness"] > 20   # the reference's gate


def test_voxel_grid_properties():
    src, _, _ = synth.make_pair(20000, 10, seed=26)
    out = oracle.voxel_grid(src, 0.2)
    assert 0 < out.shape[0] < src.shape[0]
    # every output is the mean of the inputs of one cell; cells are distinct and ordered
    inv = np.float32(1.0) / np.float32(0.2)
    ci = np.floor(src[:, :3] * inv).astype(np.int64)
    mn = ci.min(0)
    div = ci.max(0) - mn + 1
    lin = (ci[:, 0] - mn[0]) + (ci[:, 1] - mn[1]) * div[0] + (ci[:, 2] - mn[2]) * div[0] * div[1]
    cells, inverse, counts = np.unique(lin, return_inverse=True, return_counts=True)
    assert out.shape[0] == cells.shape[0]
    means = np.zeros((cells.shape[0], 3))
    np.add.at(means, inverse, src[:, :3].astype(np.float64))
    means /= counts[:, None]
    np.testing.assert_allclose(out[:, :3], means, atol=1e-4)
    assert (out[:, 3] == 1.0).all()


# ---- fixtures of the widened rows (tests/golden/make_golden_widened.py) ---------------------------------------------------
def _golden(name):
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rows_f", name))


def test_voxel_oracle_matches_golden():
    g = _golden("voxel_2k.npz")
    assert np.array_equal(oracle.voxel_grid(g["cloud"], float(g["leaf"])).view(np.uint32), g["expected"].view(np.uint32))
    assert np.array_equal(oracle.voxel_grid(g["cloud"], 0.05).view(np.uint32), g["expected_small_leaf"].view(np.uint32))


def test_map_oracle_matches_golden():
    g = _golden("map_3scans.npz")
    m = oracle.VoxelMap(float(g["resolution"]))
    for k in range(g["scans"].shape[0]):
        assert m.add_points(g["scans"][k], g["poses"][k]) == int(g["added"][k]) and len(m) == int(g["sizes"][k])
    assert np.array_equal(m.points().view(np.uint32), g["map_points"].view(np.uint32))
    nn = m.nn_cloud(g["probe"], g["probe_pose"], g["probe_pose_inv"])
    assert np.array_equal(nn.view(np.uint32), g["nn_cloud"].view(np.uint32))


def test_gicp_oracle_matches_golden():
    """The fixture comes from the NumPy restatement (tests/golden/make_golden_widened.py); the C restatement must land
    within the BASELINE tolerance of it in both of its summation modes (BFGS is chaotic in its sums: the outer iteration
    count may differ by one or two)."""
    g = _golden("gicp_1k5.npz")
    for mode in (oracle.GICP_SUMS_SEQUENTIAL, oracle.GICP_SUMS_EXACT):
        r = oracle.icp_align(g["src"], g["tgt"], oracle.default_params(method=oracle.GICP, gicp_sums=mode), want_fitness=True)
        assert r["converged"] == bool(g["converged"]) and abs(r["iterations"] - int(g["iterations"])) <= 2
        assert abs(r["n_corr"] - int(g["n_corr"])) <= 0.001 * int(g["n_corr"])
        assert np.abs(r["T"][:3, :3] - g["T"][:3, :3]).max() <= 1e-4 and np.linalg.norm(r["T"][:3, 3] - g["T"][:3, 3]) <= 1e-3
        assert abs(r["fitness"] - float(g["fitness"])) <= 1e-4 * float(g["fitness"])
    assert np.abs(oracle.gicp_covariances(g["tgt"]) - g["cov_tgt"]).max() <= 1e-9


# ---- hypothesis-driven properties of the restatement ---------------------------------------------------
from hypothesis import given, settings, strategies as st  # noqa: E402


def _cloud_pair(seed, n):
    rng = np.random.default_rng(seed)
    # a bent sheet plus a few posts: enough structure for ICP to have a unique answer at a few hundred points
    u, v = rng.uniform(-4, 4, n), rng.uniform(-4, 4, n)
    pts = np.stack([u, v, 0.15 * np.sin(u) + 0.1 * v * v / 4], 1)
    posts = rng.integers(0, n, n // 5)
    pts[posts, 2] += rng.uniform(0.2, 1.5, posts.size)
    T = synth.pose_matrix(*rng.uniform(-0.15, 0.15, 3), *rng.uniform(-0.03, 0.03, 3))
    src = np.ones((n, 4), np.float32)
    src[:, :3] = pts.astype(np.float32)
    tgt = np.ones((n, 4), np.float32)
    tgt[:, :3] = (pts @ T[:3, :3].T.astype(np.float64) + T[:3, 3] + rng.normal(0, 0.002, (n, 3))).astype(np.float32)
    return src, tgt


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(120, 400))
def test_property_permutation_invariance(seed, n):
    src, tgt = _cloud_pair(seed, n)
    rng = np.random.default_rng(seed + 1)
    a = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=8))
    b = oracle.icp_align(src[rng.permutation(n)], tgt[rng.permutation(n)], oracle.default_params(max_iterations=8))
    assert a["iterations"] == b["iterations"] and a["n_corr"] == b["n_corr"]
    assert np.abs(a["T"] - b["T"]).max() <= 1e-5


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(120, 400))
def test_property_rigid_equivariance(seed, n):
    src, tgt = _cloud_pair(seed, n)
    rng = np.random.default_rng(seed + 2)
    M = synth.pose_matrix(*rng.uniform(-5, 5, 3), *rng.uniform(-0.5, 0.5, 3)).astype(np.float64)
    mv = lambda c: np.hstack([(c[:, :3].astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32), c[:, 3:]])  # noqa: E731
    a = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=8))
    b = oracle.icp_align(mv(src), mv(tgt), oracle.default_params(max_iterations=8))
    want = M @ a["T"].astype(np.float64) @ np.linalg.inv(M)
    assert abs(a["n_corr"] - b["n_corr"]) <= max(2, n // 100)
    assert np.abs(b["T"][:3, :3] - want[:3, :3]).max() <= 1e-3 and np.linalg.norm(b["T"][:3, 3] - want[:3, 3]) <= 5e-3


@settings(max_examples=10, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(60, 300))
def test_property_identical_clouds_give_identity(seed, n):
    src, _ = _cloud_pair(seed, n)
    r = oracle.icp_align(src, src.copy(), oracle.default_params())
    assert r["converged"] and r["n_corr"] == n and np.abs(r["T"] - np.eye(4)).max() <= 1e-6 and r["mse"] <= 1e-12


# ---- GICP: two restatements, three summation orders ---------------------------------------------------------------------
@pytest.mark.parametrize("n,seed", [(1500, 1), (3000, 2), (2500, 5)])
def test_gicp_two_restatements_agree(n, seed):
    """oracle/gicp_oracle.c (hand-written kd-tree, Jacobi SVD, adjugate inverse, BFGS in C) against oracle/gicp_oracle_np.py
    (SciPy kd-tree, LAPACK SVD / inverse, BFGS in Python), both in PCL's summation order and in the exact-sum definition:
    within the BASELINE tolerance (on most pairs they even agree bit for bit), same correspondences."""
    from oracle import gicp_oracle_np as gnp
    src, tgt, _ = synth.make_pair(n, n, seed=seed)
    for sums, mode in (("sequential", oracle.GICP_SUMS_SEQUENTIAL), ("exact", oracle.GICP_SUMS_EXACT)):
        a = gnp.gicp_align(src, tgt, sums=sums)
        b = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, gicp_sums=mode))
        assert a["converged"] == b["converged"] and abs(a["iterations"] - b["iterations"]) <= 3
        assert abs(a["n_corr"] - b["n_corr"]) <= 0.001 * b["n_corr"]
        assert np.abs(a["T"][:3, :3].astype(np.float64) - b["T"][:3, :3]).max() <= 1e-4
        assert np.linalg.norm(a["T"][:3, 3].astype(np.float64) - b["T"][:3, 3]) <= 1e-3
    c_np = gnp.covariances(tgt)
    assert np.abs(c_np - oracle.gicp_covariances(tgt, pcl_order=True)).max() <= 1e-9
    assert np.abs(c_np - oracle.gicp_covariances(tgt)).max() <= 1e-9


@pytest.mark.parametrize("n,seed", [(1500, 1), (3000, 2), (2500, 5)])
def test_gicp_smooth_objective_two_restatements_agree(n, seed):
    """ORC_GICP_SUMS_SMOOTH (the inner objective without PCL's float32 rounding of the transformed points: what the GPU's QUADRATIC
    inner solver minimises, include/icpgpu.h: icpgpu_gicp_inner) in C against the same mode of the independent NumPy restatement:
    within the BASELINE tolerance, same correspondences -- and the mode stays within a few millimetres of the exact-sum definition
    (it is a different objective: PCL's rounding noise is gone, the outer loop's 1e-6 m stop then falls elsewhere)."""
    from oracle import gicp_oracle_np as gnp
    src, tgt, _ = synth.make_pair(n, n, seed=seed)
    a = gnp.gicp_align(src, tgt, sums="smooth")
    b = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, gicp_sums=oracle.GICP_SUMS_SMOOTH))
    e = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, gicp_sums=oracle.GICP_SUMS_EXACT))
    assert a["converged"] == b["converged"] and abs(a["iterations"] - b["iterations"]) <= 3
    assert abs(a["n_corr"] - b["n_corr"]) <= 0.001 * b["n_corr"]
    assert np.abs(a["T"][:3, :3].astype(np.float64) - b["T"][:3, :3]).max() <= 1e-4
    assert np.linalg.norm(a["T"][:3, 3].astype(np.float64) - b["T"][:3, 3]) <= 1e-3
    assert np.abs(b["T"][:3, :3].astype(np.float64) - e["T"][:3, :3]).max() <= 1e-3
    assert np.linalg.norm(b["T"][:3, 3].astype(np.float64) - e["T"][:3, 3]) <= 1e-2
    again = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, gicp_sums=oracle.GICP_SUMS_SMOOTH))
    assert np.array_equal(np.asarray(b["T"]), np.asarray(again["T"]))   # deterministic


def test_gicp_summation_order_sensitivity():
    """The exact-sum definition the GPU path is compared with bit for bit, PCL's sequential sums, and the sequential loop
    run backwards: 24 random pairs, every pairing within the BASELINE tolerance on at least 22 of them, and the exact
    definition no farther from PCL's order than PCL's order is from its own reversal (x3: small sample).  400 pairs (not
    in the suite, 30 s): 391 / 390 / 390 within tolerance, worst 2.4 / 3.1 / 3.1 mm."""
    from concurrent.futures import ThreadPoolExecutor

    def one(seed):
        rng = np.random.default_rng(90_000 + seed)
        n_s, n_t = int(rng.integers(3_000, 8_000)), int(rng.integers(3_000, 8_000))
        gate = float(rng.choice([0.5, 1.0, 2.0]))
        src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
        r = [oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_correspondence_distance=gate, gicp_sums=m))
             for m in (oracle.GICP_SUMS_EXACT, oracle.GICP_SUMS_SEQUENTIAL, oracle.GICP_SUMS_SEQUENTIAL_REVERSED)]

        def d(a, b):
            return (float(np.abs(a["T"][:3, :3].astype(np.float64) - b["T"][:3, :3]).max()),
                    float(np.linalg.norm(a["T"][:3, 3].astype(np.float64) - b["T"][:3, 3])))
        return d(r[0], r[1]), d(r[1], r[2])

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, range(24)))
    for k in (0, 1):
        assert sum(dR <= 1e-4 and dt <= 1e-3 for dR, dt in (r[k] for r in res)) >= 22
    worst_exact = max(r[0][1] for r in res)
    worst_yard = max(r[1][1] for r in res)
    assert worst_exact <= max(3.0 * worst_yard, 1e-3), (worst_exact, worst_yard)
