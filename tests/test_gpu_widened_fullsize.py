"""The widened rows of SURVEY.md 8(f) against the oracle AT THE SIZES BASELINE.json NAMES ("next round" item 1):

(a) f1, the solver the reference instantiates (pcl::GeneralizedIterativeClosestPoint, /root/reference/src/icpslam/icp_odometer.cpp:188-201)
    on the 200k x 200k headline pair: covariances, iterations, correspondences and the transform bit-identical to the oracle's
    exact-sum restatement, through the host loop AND the resident device solver;
(b) f4, the mapper's flow (/root/reference/src/icpslam/octree_mapper.cpp:101-172) at config 3's scale: 200k-point scans grow a map
    to more than a million points, the map and the nn cloud bit-identical to oracle/map_oracle.c, the 30-iteration GICP refinement
    (octree_mapper.h:56) against the oracle on that nn cloud;
(c) the opt-in QUADRATIC inner solver on a voxel-filtered drive of the reference's pipeline: the odometer's accept gate decides the
    same for every scan, and the per-scan distance to the exact mode stays where the round-5 campaign measured it.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from icpslam_amd import GICP, GICP_INNER_EXACT, GICP_INNER_QUADRATIC, synth
from icpslam_amd.mapper import OctreeMapper
from icpslam_amd.sequence import pose_from_matrix, pose_inverse, pose_to_matrix

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---- (a) GICP at 200k x 200k ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def pair200k():
    return synth.make_pair(200000, 200000, seed=4)


@pytest.fixture(scope="module")
def gicp_ref_200k(pair200k):
    src, tgt, _ = pair200k
    return oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_iterations=10), want_fitness=True)


def test_gicp_covariances_200k_equal_the_oracle(ctx, pair200k):
    src, tgt, _ = pair200k
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    ctx.set_source(src)
    ctx.set_target(tgt)
    for of_target, cloud in ((False, src), (True, tgt)):
        got = ctx.gicp_covariances(of_target)
        ref = oracle.gicp_covariances(cloud)
        assert got.shape == ref.shape == (200000, 3, 3)
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_gicp_headline_pair_200k_bit_identical_to_the_oracle(ctx, pair200k, gicp_ref_200k):
    """The default path of a lone context (ICPGPU_GICP_DEVICE=auto: the host loop over the evaluation server or the device solver,
    whichever the context settled on -- same bits either way)."""
    src, tgt, _ = pair200k
    ref = gicp_ref_200k
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    ctx.set_source(src)
    ctx.set_target(tgt)
    got = ctx.align(want_fitness=True)
    assert (got["converged"], got["iterations"], got["n_corr"]) == (ref["converged"], ref["iterations"], ref["n_corr"])
    assert np.array_equal(_bits(got["T"]), _bits(np.asarray(ref["T"], np.float32)))
    assert abs(got["fitness"] - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    assert ref["iterations"] <= 10 and ref["n_corr"] > 190000


@pytest.mark.parametrize("device", ["0", "1"])
def test_gicp_headline_pair_200k_host_loop_and_device_solver(built, gicp_ref_200k, device):
    """ICPGPU_GICP_DEVICE=0 (BFGS on the host over the resident evaluation server) and =1 (the whole BFGS in gicp_solve_kernel) on
    the 200k x 200k pair: both land on the oracle's bits.  The switch is read once per process, hence sub-processes."""
    code = (
        "import numpy as np\n"
        "from icpslam_amd import Context, GICP, synth\n"
        "src, tgt, _ = synth.make_pair(200000, 200000, seed=4)\n"
        "with Context(0) as ctx:\n"
        "    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)\n"
        "    ctx.set_source(src); ctx.set_target(tgt)\n"
        "    r = ctx.align(want_fitness=True)\n"
        "    p = ctx.profile()\n"
        "print(r['T'].tobytes().hex(), int(r['converged']), r['iterations'], r['n_corr'], float(r['fitness']).hex(), p.gicp_host_solves, p.gicp_device_solves)\n")
    env = dict(os.environ, ICPGPU_GICP_DEVICE=device, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    t_hex, conv, iters, n_corr, fit_hex, host_solves, dev_solves = res.stdout.strip().splitlines()[-1].split()
    ref = gicp_ref_200k
    assert (int(conv), int(iters), int(n_corr)) == (int(ref["converged"]), ref["iterations"], ref["n_corr"])
    assert bytes.fromhex(t_hex) == np.asarray(ref["T"], np.float32).tobytes()
    assert abs(float.fromhex(fit_hex) - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    if device == "0":
        assert int(host_solves) == ref["iterations"] and int(dev_solves) == 0
    else:
        assert int(dev_solves) == ref["iterations"] and int(host_solves) == 0


# ---- (b) the mapper's flow at config 3's scale -----------------------------------------------------------------------------------
def test_mapper_flow_200k_scans_into_a_million_point_map(ctx):
    """octree_mapper.cpp:133-172 with 200k-point scans and a resolution (2 cm) at which seven of them grow the map past a million
    points -- BASELINE config 3's "200k scan vs 1M-pt local Octree submap" through icpgpu_map_* instead of a prepared target.
    Scans 1..6 are refined against the map with the reference's 30-iteration GICP and then grow it; the CPU side repeats every
    step: the map (bit for bit, after every scan), the nn cloud (bit for bit), the refinement (bit-identical transform, equal
    iteration and correspondence counts) -- then one more scan against the million-point map."""
    scene = synth.make_scene(seed=41)
    poses = [synth.pose_matrix(1.5 * k, 0.1 * k, 0.0, 0.0, 0.0, 0.03 * k) for k in range(8)]
    with ThreadPoolExecutor(8) as ex:
        scans = list(ex.map(lambda kp: synth.scan(scene, kp[1], 200000, seed=410 + kp[0]), enumerate(poses)))
    err = synth.pose_matrix(0.12, -0.08, 0.02, 0.0, 0.0, 0.008)        # the raw odometry's error the refinement has to undo
    res = 0.02
    mapper = OctreeMapper(ctx, octree_resolution=res)                   # GICP, 30 iterations: octree_mapper.cpp:104, octree_mapper.h:56
    ref = oracle.VoxelMap(res)
    p_gicp = oracle.default_params(method=oracle.GICP, max_iterations=30)
    refined_ok = 0
    for k, (scan, P) in enumerate(zip(scans, poses)):
        raw = pose_from_matrix((P.astype(np.float64) @ err.astype(np.float64)).astype(np.float32) if k else P)
        raw_M, raw_Minv = pose_to_matrix(raw), pose_to_matrix(pose_inverse(raw))
        if k == 0:
            ok, transform, refined, info = mapper.refineTransformAndGrowMap(scan, raw)
            assert not ok and info["seeded"] and info["added"] == ref.add_points(scan, raw_M)
            continue
        if k == 7:
            assert mapper.map_size >= 1_000_000          # the last scan meets a million-point map
        # approxNearestNeighbors: the nn cloud itself, fetched, against the oracle's
        nn_gpu = mapper.approxNearestNeighbors(scan, raw, want_cloud=True)
        nn_ref = ref.nn_cloud(scan, raw_M, raw_Minv)
        assert nn_gpu.shape == nn_ref.shape == (200000, 4)
        assert np.array_equal(_bits(nn_gpu), _bits(nn_ref)), k
        # estimateTransformICP on it
        ok, transform, res_gpu = mapper.estimateTransformICP()
        o = oracle.icp_align(scan, nn_ref, p_gicp)
        assert ok == bool(o["converged"]) and ok
        assert (res_gpu["iterations"], res_gpu["n_corr"]) == (o["iterations"], o["n_corr"]), k
        assert np.array_equal(_bits(res_gpu["T"]), _bits(np.asarray(o["T"], np.float32))), k
        # grow both maps with the refined pose
        from icpslam_amd.sequence import pose_compose
        refined = pose_compose(raw, transform)
        added = ctx.map_add_source(pose_to_matrix(refined))
        assert added == ref.add_points(scan, pose_to_matrix(refined)), k
        assert mapper.map_size == len(ref)
        resid = np.linalg.inv(P.astype(np.float64)) @ pose_to_matrix(refined).astype(np.float64)
        # every refinement ends nearer to the truth than the raw odometry was (14.6 cm off); the map's own drift (it is built from
        # refined poses) keeps the later ones from doing better than ~11 cm (the CPU flow: 1.7, 3.3, 5.0, 6.8, 9.1, 11.1, 11.3 cm)
        refined_ok += np.linalg.norm(resid[:3, 3]) < np.linalg.norm(err[:3, 3])
        if k in (3, 7):
            assert np.array_equal(_bits(mapper.map_cloud()), _bits(ref.points())), k
    assert mapper.map_size == len(ref) >= 1_000_000, len(ref)
    assert refined_ok == 7, refined_ok


# ---- (c) the QUADRATIC inner solver on a voxel-filtered drive ---------------------------------------------------------------------
def test_quadratic_mode_on_a_voxel_filtered_drive_keeps_the_gate_decisions(built):
    """The reference's per-scan pipeline (VoxelGrid 0.2 m + GICP + the accept gate `converged and fitness < 20`,
    icp_odometer.cpp:177,188-201) over a 33-scan drive of raw 200k-point scans, once per inner solver, every scan registered
    as the odometer does it (the target is the last ACCEPTED scan, icp_odometer.cpp:201-209).  Scan 15 is corrupted (moved 500 m up:
    no correspondence within the 1 m gate -> not converged -> rejected, and scan 16 registers against scan 14).  The modes must
    take the same decision on every scan; per scan they differ by what PCL's float32 rounding of the
    transformed points contributes -- round 5 measured a median of 1.2 mm, 3.4 mm at the 90th percentile and 9.9 mm at worst over
    120 such pairs (profiles/r05_gicp_quadratic.txt) -- and 0.90 / 2.65 / 5.27 mm on this drive (round 6) -- asserted here: median <= 2.5 mm, 90th
    percentile <= 5 mm, nothing beyond 1.2 cm, rotations within 1e-4."""
    from icpslam_amd import Context
    n_scan = 33
    rng = np.random.default_rng(5)
    scene = synth.make_scene(5, extent=120.0)
    poses = [np.eye(4)]
    for k in range(n_scan - 1):
        step = synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3)))
        poses.append(poses[-1] @ step)
    with ThreadPoolExecutor(8) as ex:
        scans = list(ex.map(lambda kp: synth.scan(scene, kp[1], 200000, seed=7000 + kp[0]), enumerate(poses)))
    scans[15] = scans[15].copy()
    scans[15][:, 2] += 500.0
    gate = lambda r: bool(r["converged"]) and r["fitness"] < 20.0
    runs = {}
    for inner in (GICP_INNER_EXACT, GICP_INNER_QUADRATIC):
        with Context(0) as c:
            c.set_params(c.default_params(), method=GICP, max_iterations=10, gicp_inner=inner)
            c.set_source_voxel_filtered(scans[0], 0.2)
            c.promote_source_to_target()
            out = []
            for k in range(1, n_scan):
                n_f = c.set_source_voxel_filtered(scans[k], 0.2)
                assert 5000 < n_f < 60000
                out.append(c.align(want_fitness=True))
                if gate(out[-1]):
                    c.promote_source_to_target()       # *prev_cloud_ = *curr_cloud_, icp_odometer.cpp:209
            if inner == GICP_INNER_QUADRATIC:
                assert c.profile().gicp_quadratic_solves > 0
        runs[inner] = out
    ex_, qu = runs[GICP_INNER_EXACT], runs[GICP_INNER_QUADRATIC]
    assert [gate(r) for r in qu] == [gate(r) for r in ex_]
    assert not gate(ex_[14]) and sum(gate(r) for r in ex_) == n_scan - 2     # (scan 15 = registration number 14)
    ok = [k for k in range(n_scan - 1) if gate(ex_[k])]
    dt = np.array([np.linalg.norm(qu[k]["T"][:3, 3].astype(np.float64) - ex_[k]["T"][:3, 3]) for k in ok])
    dR = np.array([np.abs(qu[k]["T"][:3, :3].astype(np.float64) - ex_[k]["T"][:3, :3]).max() for k in ok])
    print(f"quadratic vs exact over {len(ok)} accepted pipeline scans: dt median {np.median(dt) * 1e3:.2f} mm, 90th {np.quantile(dt, .9) * 1e3:.2f} mm, "
          f"worst {dt.max() * 1e3:.2f} mm; worst dR {dR.max():.1e}")
    assert np.median(dt) <= 2.5e-3 and np.quantile(dt, 0.9) <= 5e-3 and dt.max() <= 1.2e-2, (np.median(dt), np.quantile(dt, 0.9), dt.max())
    assert dR.max() <= 1e-4
    # the first pairs on the CPU: the exact mode lands on the oracle's bits; the quadratic mode stays as near to the oracle's
    # restatement of ITS objective (GICP_SUMS_SMOOTH) as to the exact mode -- on these pairs BFGS amplifies the last bits of either
    # (round 6, first GPU run: 3.1 mm on pair 1; tests/test_gpu_gicp_quadratic.py holds the statistics over random pairs)
    for k in (1, 2):
        s = oracle.voxel_grid(scans[k], 0.2)
        t = oracle.voxel_grid(scans[k - 1], 0.2)
        e = oracle.icp_align(s, t, oracle.default_params(method=oracle.GICP, max_iterations=10), want_fitness=True)
        assert np.array_equal(_bits(ex_[k - 1]["T"]), _bits(np.asarray(e["T"], np.float32)))
        assert (ex_[k - 1]["iterations"], ex_[k - 1]["n_corr"]) == (e["iterations"], e["n_corr"])
        assert abs(ex_[k - 1]["fitness"] - e["fitness"]) <= 1e-9 * max(1.0, e["fitness"])
        o = oracle.icp_align(s, t, oracle.default_params(method=oracle.GICP, max_iterations=10, gicp_sums=oracle.GICP_SUMS_SMOOTH))
        assert np.abs(qu[k - 1]["T"][:3, :3] - o["T"][:3, :3]).max() <= 1e-4
        assert np.linalg.norm(qu[k - 1]["T"][:3, 3] - o["T"][:3, 3]) <= 1.2e-2
