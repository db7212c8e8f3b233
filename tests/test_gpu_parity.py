"""GPU parity tests: HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): NN indices / squared distances / correspondence counts bit-exact;
rigid transform within 1e-4 (R, max-abs element) and 1e-3 m (t) of the oracle.
"""
import numpy as np
import pytest

import oracle
from icpslam_amd import synth

pytestmark = pytest.mark.gpu

R_TOL = 1e-4   # max |R_gpu - R_oracle| element
T_TOL = 1e-3   # |t_gpu - t_oracle| in metres


def _dR(a, b):
    return float(np.abs(a[:3, :3].astype(np.float64) - b[:3, :3].astype(np.float64)).max())


def _dt(a, b):
    return float(np.linalg.norm(a[:3, 3].astype(np.float64) - b[:3, 3].astype(np.float64)))


def _rand_T(rng, scale=1.0):
    return synth.pose_matrix(*(rng.uniform(-0.5, 0.5, 3) * scale), *(np.deg2rad(rng.uniform(-3, 3, 3)) * scale))


# ---------------------------------------------------------------------------------------------------------
# a2: nearest-neighbour search -- bit exact
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_s,n_t,seed", [(1, 1, 0), (7, 3, 1), (64, 1000, 2), (257, 1025, 3), (1000, 8191, 4),
                                            (5000, 5000, 5), (4097, 20000, 6), (20000, 333, 7)])
def test_nn_bit_exact_random(ctx, n_s, n_t, seed):
    rng = np.random.default_rng(seed)
    src = np.ones((n_s, 4), np.float32)
    tgt = np.ones((n_t, 4), np.float32)
    src[:, :3] = rng.uniform(-40, 40, (n_s, 3))
    tgt[:, :3] = rng.uniform(-40, 40, (n_t, 3))
    T = _rand_T(rng)
    ctx.set_source(src)
    ctx.set_target(tgt)
    idx, d2 = ctx.nn(T)
    ridx, rd2 = oracle.nn(src, tgt, T, nn_mode=oracle.NN_BRUTE)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))


def test_nn_bit_exact_scan_kdtree_oracle(ctx):
    src, tgt, _ = synth.make_pair(30000, 30000, seed=21)
    ctx.set_source(src)
    ctx.set_target(tgt)
    idx, d2 = ctx.nn(np.eye(4))
    ridx, rd2 = oracle.nn(src, tgt, np.eye(4), nn_mode=oracle.NN_KDTREE)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))


def test_nn_ties_pick_lowest_index(ctx):
    # integer lattice: many exact ties; duplicates of every target point at higher indices
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
    tgt = np.ones((g.shape[0] * 2, 4), np.float32)
    tgt[: g.shape[0], :3] = g
    tgt[g.shape[0]:, :3] = g
    src = np.ones((300, 4), np.float32)
    src[:, :3] = np.random.default_rng(0).integers(0, 8, (300, 3)) + 0.5    # equidistant to 8 lattice points
    ctx.set_source(src)
    ctx.set_target(tgt)
    idx, d2 = ctx.nn(np.eye(4))
    ridx, rd2 = oracle.nn(src, tgt, np.eye(4), nn_mode=oracle.NN_BRUTE)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)
    assert (idx < g.shape[0]).all()


def test_nn_empty_and_nonfinite(ctx):
    src = np.ones((10, 4), np.float32)
    src[:, :3] = np.arange(30).reshape(10, 3)
    src[3, 0] = np.nan
    src[5, 1] = np.inf
    tgt = np.ones((100, 4), np.float32)
    tgt[:, :3] = np.random.default_rng(1).uniform(0, 30, (100, 3))
    ctx.set_source(src)
    ctx.set_target(tgt)
    idx, d2 = ctx.nn(np.eye(4))
    assert idx[3] == -1 and np.isinf(d2[3])
    ok = np.ones(10, bool)
    ok[[3, 5]] = False
    ridx, rd2 = oracle.nn(src, tgt, np.eye(4), nn_mode=oracle.NN_BRUTE)
    assert np.array_equal(idx[ok], ridx[ok]) and np.array_equal(d2[ok], rd2[ok])
    # empty target: every key empty
    ctx.set_target(np.zeros((0, 4), np.float32))
    idx, d2 = ctx.nn(np.eye(4))
    assert (idx == -1).all() and np.isinf(d2).all()


# ---------------------------------------------------------------------------------------------------------
# a3 + a4: rejection + reduction; a5: solve; a6: transform
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,seed,r", [(5000, 31, 1.0), (20000, 32, 0.3), (777, 33, 5.0)])
def test_reduce_matches_oracle(ctx, n, seed, r):
    src, tgt, _ = synth.make_pair(n, n, seed=seed)
    T = _rand_T(np.random.default_rng(seed), 0.2)
    ctx.set_source(src)
    ctx.set_target(tgt)
    idx, d2 = ctx.nn(T)
    sums = ctx.reduce(T, r)
    ref = oracle.reduce(src, tgt, T, idx, d2, r)
    assert sums[0] == ref[0]                        # accepted-pair count: exact
    np.testing.assert_allclose(sums, ref, rtol=1e-12, atol=1e-9)
    Tk = ctx.solve(sums)
    np.testing.assert_allclose(Tk, oracle.umeyama(ref), rtol=0, atol=1e-10)


def test_transform_bit_exact(ctx):
    src, _, T = synth.make_pair(12345, 10, seed=41)
    ctx.set_source(src)
    out = ctx.transform(T)
    ref = oracle.transform_cloud(src, T)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


# ---------------------------------------------------------------------------------------------------------
# a1..a10: full align
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_s,n_t,seed,iters", [(5000, 5000, 1, 10), (5000, 5000, 1, 30), (3000, 7000, 51, 30),
                                                  (20000, 20000, 52, 10), (50000, 50000, 2, 30)])
def test_align_matches_oracle(ctx, n_s, n_t, seed, iters):
    src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
    ctx.set_params(ctx.default_params(), max_iterations=iters)
    ctx.set_source(src)
    ctx.set_target(tgt)
    got = ctx.align(want_cloud=True, want_fitness=True)
    ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=iters), want_cloud=True, want_fitness=True)
    assert got["converged"] == ref["converged"]
    assert got["iterations"] == ref["iterations"]
    assert got["state"] == ref["state"]
    assert got["n_corr"] == ref["n_corr"]
    assert _dR(got["T"], ref["T"]) <= R_TOL
    assert _dt(got["T"], ref["T"]) <= T_TOL
    assert abs(got["mse"] - ref["mse"]) <= 1e-9 * max(1.0, ref["mse"])
    assert abs(got["fitness"] - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    np.testing.assert_allclose(got["cloud"], ref["cloud"], rtol=0, atol=1e-4)


def test_align_known_answer(ctx):
    src, tgt, T_gt = synth.make_known_answer_pair(8000, seed=61)
    ctx.set_params(ctx.default_params(), max_iterations=50)
    ctx.set_source(src)
    ctx.set_target(tgt)
    got = ctx.align()
    assert got["converged"]
    assert _dR(got["T"], T_gt) <= 1e-5
    assert _dt(got["T"], T_gt) <= 1e-4


def test_align_with_guess_and_forced_iterations(ctx):
    src, tgt, T_gt = synth.make_pair(6000, 6000, seed=71)
    guess = T_gt.copy()
    guess[:3, 3] += 0.05
    ctx.set_params(ctx.default_params(), max_iterations=7, force_iterations=1)
    ctx.set_source(src)
    ctx.set_target(tgt)
    got = ctx.align(guess=guess)
    ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=7, force_iterations=1), guess=guess)
    assert got["iterations"] == ref["iterations"] == 7
    assert got["n_corr"] == ref["n_corr"]
    assert _dR(got["T"], ref["T"]) <= R_TOL and _dt(got["T"], ref["T"]) <= T_TOL


def test_align_degenerate_inputs(ctx):
    src, tgt, _ = synth.make_pair(2000, 2000, seed=81)
    ctx.set_params(ctx.default_params())
    # empty target: PCL refuses it, align returns converged = false, T = I
    ctx.set_source(src)
    ctx.set_target(np.zeros((0, 4), np.float32))
    got = ctx.align(want_cloud=True)
    assert not got["converged"] and got["iterations"] == 0
    assert np.array_equal(got["T"], np.eye(4, dtype=np.float32))
    # empty source: no correspondences
    ctx.set_source(np.zeros((0, 4), np.float32))
    ctx.set_target(tgt)
    got = ctx.align()
    ref = oracle.icp_align(np.zeros((0, 4), np.float32), tgt)
    assert not got["converged"] and got["state"] == ref["state"] == 5
    # clouds farther apart than the correspondence gate: < 3 correspondences
    far = src.copy()
    far[:, 0] += 500.0
    ctx.set_source(far)
    got = ctx.align()
    ref = oracle.icp_align(far, tgt)
    assert not got["converged"] and got["state"] == ref["state"] == 5 and got["n_corr"] == ref["n_corr"]
    # a NaN point is simply never matched
    bad = src.copy()
    bad[17, :3] = np.nan
    ctx.set_source(bad)
    got = ctx.align()
    good = np.delete(src, 17, axis=0)
    ref = oracle.icp_align(good, tgt)
    assert got["converged"] == ref["converged"] and got["n_corr"] == ref["n_corr"]
    assert _dR(got["T"], ref["T"]) <= R_TOL and _dt(got["T"], ref["T"]) <= T_TOL


def test_pcl_shaped_front_end(built):
    """Reads like the reference's call site (icp_odometer.cpp:188-201)."""
    from icpslam_amd import IterativeClosestPoint
    curr, prev, _ = synth.make_pair(5000, 5000, seed=1)
    icp = IterativeClosestPoint()
    icp.setMaximumIterations(10.0)           # the reference passes a `const double`
    icp.setTransformationEpsilon(1e-6)
    icp.setMaxCorrespondenceDistance(1.0)
    icp.setRANSACIterations(0)
    icp.setInputSource(curr)
    icp.setInputTarget(prev)
    aligned = icp.align()
    T = icp.getFinalTransformation().astype(np.float64)
    ref = oracle.icp_align(curr, prev, oracle.default_params(max_iterations=10), want_fitness=True)
    assert icp.hasConverged() and icp.getFitnessScore() < 20
    assert aligned.shape == curr.shape
    assert _dR(T, ref["T"]) <= R_TOL and _dt(T, ref["T"]) <= T_TOL
    assert abs(icp.getFitnessScore() - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])


def test_align_batch_matches_single_aligns(ctx):
    """BASELINE config 4 in miniature: independent pairs through icpgpu_align_batch == one icpgpu_align per pair."""
    pairs = [synth.make_pair(6000 + 500 * k, 7000, seed=300 + k)[:2] for k in range(7)]
    pairs.append((np.zeros((0, 4), np.float32), pairs[0][1]))          # empty source inside a batch
    ctx.set_params(ctx.default_params(), max_iterations=10)
    batch = ctx.align_batch([p[0] for p in pairs], [p[1] for p in pairs], want_fitness=True)
    assert len(batch) == len(pairs)
    for (s, t), got in zip(pairs, batch):
        ctx.set_source(s)
        ctx.set_target(t)
        one = ctx.align(want_fitness=True)
        assert (got["converged"], got["iterations"], got["state"], got["n_corr"]) == \
               (one["converged"], one["iterations"], one["state"], one["n_corr"])
        assert np.array_equal(got["T"], one["T"])
        assert got["fitness"] == one["fitness"] or (np.isnan(got["fitness"]) and np.isnan(one["fitness"]))
    ref = oracle.icp_align(pairs[3][0], pairs[3][1], oracle.default_params(max_iterations=10))
    assert _dR(batch[3]["T"], ref["T"]) <= R_TOL and _dt(batch[3]["T"], ref["T"]) <= T_TOL


# ---- BASELINE config 4 at its workload: icpgpu_align_batch on 50k-point pairs -------------------------------------------------
def _oracle_many(pairs, params):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:   # the oracle releases the GIL (ctypes)
        return list(ex.map(lambda p: oracle.icp_align(p[0], p[1], params, want_fitness=True), pairs))


@pytest.fixture(scope="module")
def pairs50k():
    return [synth.make_pair(50000, 50000, seed=1000 + k)[:2] for k in range(16)]   # seeds of SURVEY.md 8(d) C4


def test_align_batch_16_pairs_of_50k_match_oracle(ctx, pairs50k):
    """16 of config 4's 512 pairs: 50k points -> nn_quad_kernel with the previous-neighbour bound, several worker contexts in
    flight per host thread (the round-robin scheduler of icpgpu_align_batch), against the oracle pair by pair."""
    ctx.set_params(ctx.default_params(), max_iterations=10)
    got = ctx.align_batch([p[0] for p in pairs50k], [p[1] for p in pairs50k], want_fitness=True)
    ref = _oracle_many(pairs50k, oracle.default_params(max_iterations=10))
    for g, r in zip(got, ref):
        assert (g["converged"], g["iterations"], g["state"], g["n_corr"]) == (r["converged"], r["iterations"], r["state"], r["n_corr"])
        assert _dR(g["T"], r["T"]) <= R_TOL and _dt(g["T"], r["T"]) <= T_TOL
        assert abs(g["fitness"] - r["fitness"]) <= 1e-9 * max(1.0, r["fitness"])
    # every scheduler shape gives the same bits (threads x contexts per thread: 1 x 8, 2 x 4, 8 x 1)
    import os
    for t, k in ((1, 8), (2, 4), (8, 1)):
        os.environ["ICPGPU_BATCH_THREADS"], os.environ["ICPGPU_BATCH_DEPTH"] = str(t), str(k)
        try:
            again = ctx.align_batch([p[0] for p in pairs50k], [p[1] for p in pairs50k], want_fitness=True)
        finally:
            del os.environ["ICPGPU_BATCH_THREADS"], os.environ["ICPGPU_BATCH_DEPTH"]
        for a, b in zip(again, got):
            assert np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"] and a["fitness"] == b["fitness"]


def test_align_batch_gicp_8_pairs_of_50k_match_oracle(ctx, pairs50k):
    """The same through the solver the reference instantiates (GICP): bit-identical to the exact-sum oracle, pair by pair."""
    from icpslam_amd import GICP
    pairs = pairs50k[:8]
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    got = ctx.align_batch([p[0] for p in pairs], [p[1] for p in pairs], want_fitness=True)
    ref = _oracle_many(pairs, oracle.default_params(method=oracle.GICP, max_iterations=10))
    for g, r in zip(got, ref):
        assert (g["converged"], g["iterations"], g["n_corr"]) == (r["converged"], r["iterations"], r["n_corr"])
        assert np.array_equal(g["T"].view(np.uint32), np.asarray(r["T"], np.float32).view(np.uint32))
        assert abs(g["fitness"] - r["fitness"]) <= 1e-9 * max(1.0, r["fitness"])
    ctx.set_params(ctx.default_params())


@pytest.mark.parametrize("seed", range(10))
def test_align_campaign_slice(ctx, seed):
    """A 10-pair slice of scripts/align_campaign.py (300 pairs in round 1): random sizes 33k-60k x 20k-80k, gates 0.3-2 m,
    5-30 iterations; iterations, correspondences and state equal to the oracle's, transform within the BASELINE tolerance."""
    rng = np.random.default_rng(50_000 + seed)
    n_s, n_t = int(rng.integers(33_000, 60_000)), int(rng.integers(20_000, 80_000))
    gate = float(rng.choice([0.3, 1.0, 2.0]))
    iters = int(rng.choice([5, 10, 30]))
    src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
    ctx.set_params(ctx.default_params(), max_iterations=iters, max_correspondence_distance=gate)
    ctx.set_source(src)
    ctx.set_target(tgt)
    got = ctx.align(want_fitness=True)
    ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=iters, max_correspondence_distance=gate), want_fitness=True)
    assert (got["converged"], got["iterations"], got["state"], got["n_corr"]) == (ref["converged"], ref["iterations"], ref["state"], ref["n_corr"])
    assert _dR(got["T"], ref["T"]) <= R_TOL and _dt(got["T"], ref["T"]) <= T_TOL
    assert abs(got["fitness"] - ref["fitness"]) <= 1e-6 * max(1.0, ref["fitness"])
    ctx.set_params(ctx.default_params())


def test_registration_mirrors_keep_their_pcl_semantics(built):
    """icpslam_amd.GeneralizedIterativeClosestPoint (what the reference instantiates) runs GICP, IterativeClosestPoint runs
    point-to-point ICP, and two objects sharing the cached context do not disturb each other's getFitnessScore()."""
    from icpslam_amd import GeneralizedIterativeClosestPoint, IterativeClosestPoint
    curr, prev, _ = synth.make_pair(6000, 6000, seed=61)
    a = GeneralizedIterativeClosestPoint()
    a.setMaximumIterations(10); a.setTransformationEpsilon(1e-6); a.setMaxCorrespondenceDistance(1.0); a.setRANSACIterations(0)
    a.setInputSource(curr); a.setInputTarget(prev)
    a.align()
    fit_a = a.getFitnessScore()
    ref = oracle.icp_align(curr, prev, oracle.default_params(method=oracle.GICP, max_iterations=10), want_fitness=True)
    assert a.hasConverged() == ref["converged"] and a.result["iterations"] == ref["iterations"]
    assert np.array_equal(a.getFinalTransformation().view(np.uint32), np.asarray(ref["T"], np.float32).view(np.uint32))
    assert abs(fit_a - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    b = IterativeClosestPoint()                       # same device: the same cached context
    b.setMaximumIterations(3); b.setInputSource(prev); b.setInputTarget(curr)
    b.align()
    refb = oracle.icp_align(prev, curr, oracle.default_params(max_iterations=3), want_fitness=True)
    assert b.result["iterations"] == refb["iterations"] and _dt(b.getFinalTransformation(), refb["T"]) <= T_TOL
    assert abs(a.getFitnessScore() - fit_a) <= 1e-9 * max(1.0, fit_a)          # still A's, after B used the context
    assert abs(b.getFitnessScore() - refb["fitness"]) <= 1e-9 * max(1.0, refb["fitness"])
