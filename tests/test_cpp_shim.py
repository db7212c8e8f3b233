"""The C++ shim (include/icpgpu_registration.hpp) driven exactly like the reference's call site."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from icpslam_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_demo(tmp_path, name="shim_demo"):
    exe = tmp_path / name
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", str(exe), "-L", libdir, "-licpgpu",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return exe


def _run(exe, tmp_path, src, tgt, iters, *extra):
    a, b = tmp_path / "src.bin", tmp_path / "tgt.bin"
    src.tofile(a)
    tgt.tofile(b)
    return subprocess.run([str(exe), str(a), str(src.shape[0]), str(b), str(tgt.shape[0]), str(iters), *extra],
                          capture_output=True, text=True)


def test_shim_compiles_and_fails_loudly_without_gpu(built, tmp_path):
    import torch
    exe = _build_demo(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    src, tgt, _ = synth.make_pair(100, 100, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_shim_matches_oracle(built, tmp_path):
    exe = _build_demo(tmp_path)
    src, tgt, _ = synth.make_pair(5000, 5000, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 0, r.stderr
    vals = r.stdout.split()
    ok, iters, fit = int(vals[0]), int(vals[1]), float(vals[2])
    T = np.array([float(v) for v in vals[3:19]]).reshape(4, 4).T
    ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=10), want_fitness=True, want_cloud=True)
    assert ok == 1 and iters == ref["iterations"]
    assert np.abs(T[:3, :3] - ref["T"][:3, :3]).max() <= 1e-4 and np.linalg.norm(T[:3, 3] - ref["T"][:3, 3]) <= 1e-3
    assert abs(fit - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    c = ref["cloud"].astype(np.float64)
    assert abs(float(vals[19]) - (c[:, 0] + 2 * c[:, 1] + 3 * c[:, 2] + c[:, 3]).sum()) <= 1e-2


def test_gicp_shim_compiles_and_fails_loudly_without_gpu(built, tmp_path):
    import torch
    exe = _build_demo(tmp_path, "gicp_shim_demo")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    src, tgt, _ = synth.make_pair(100, 100, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_gicp_shim_matches_oracle_and_objects_do_not_disturb_each_other(built, tmp_path):
    """icpgpu::GeneralizedIterativeClosestPoint -- the type the reference's call site swaps to -- runs
    GICP; align() leaves width = size, height = 1, is_dense on clouds that have those members; a second registration object
    on the same thread (shared cached context) does not change what the first one's getFitnessScore() returns."""
    exe = _build_demo(tmp_path, "gicp_shim_demo")
    src, tgt, _ = synth.make_pair(5000, 5000, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 0, r.stderr
    l1, l2 = [ln.split() for ln in r.stdout.strip().splitlines()]
    ref = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_iterations=10), want_fitness=True)
    T = np.array([float(v) for v in l1[3:19]]).reshape(4, 4).T
    assert int(l1[0]) == int(ref["converged"]) and int(l1[1]) == ref["iterations"]
    assert np.abs(T[:3, :3] - ref["T"][:3, :3]).max() <= 1e-4 and np.linalg.norm(T[:3, 3] - ref["T"][:3, 3]) <= 1e-3
    assert abs(float(l1[2]) - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    assert (int(l1[19]), int(l1[20]), int(l1[21])) == (src.shape[0], 1, 1)
    assert abs(float(l2[0]) - float(l1[2])) <= 1e-9 * max(1.0, float(l1[2]))          # A's fitness, after B used the context
    other = oracle.icp_align(tgt, src, oracle.default_params(max_iterations=3), want_fitness=True)
    assert abs(float(l2[1]) - other["fitness"]) <= 1e-9 * max(1.0, other["fitness"])


def _run_map(exe, tmp_path, scan0, scan1, pose, pose_inv):
    a, b = tmp_path / "s0.bin", tmp_path / "s1.bin"
    scan0.tofile(a)
    scan1.tofile(b)
    cm = lambda M: [repr(float(v)) for v in np.asarray(M, np.float32).T.reshape(-1)]  # column-major
    return subprocess.run([str(exe), str(a), str(scan0.shape[0]), str(b), str(scan1.shape[0])] + cm(pose) + cm(pose_inv),
                          capture_output=True, text=True)


def test_map_shim_compiles_and_fails_loudly_without_gpu(built, tmp_path):
    import torch
    exe = _build_demo(tmp_path, "map_demo")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    s0, s1, T = synth.make_pair(100, 100, seed=1)
    r = _run_map(exe, tmp_path, s0, s1, T, np.linalg.inv(T))
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_map_shim_matches_oracle(built, tmp_path):
    """octree_mapper.cpp:133-172 through the C++ shim (OctreeMap + IterativeClosestPoint::setInputTargetFromMap)."""
    exe = _build_demo(tmp_path, "map_demo")
    scan1, scan0, Tgt = synth.make_pair(20000, 20000, seed=7)      # scan1 (source) sits at Tgt in scan0's frame
    pose = Tgt.astype(np.float64).copy()
    pose[:3, 3] += (0.1, -0.05, 0.0)                                # raw odometry with an error
    pose = pose.astype(np.float32)
    pose_inv = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    r = _run_map(exe, tmp_path, scan0, scan1, pose, pose_inv)
    assert r.returncode == 0, r.stderr
    v = r.stdout.split()
    ref = oracle.VoxelMap(0.5)
    ref.add_points(scan0, np.eye(4))
    assert int(v[0]) == len(ref)
    nn = ref.nn_cloud(scan1, pose, pose_inv)
    assert int(v[1]) == nn.shape[0]
    o = oracle.icp_align(scan1, nn, oracle.default_params(max_iterations=30))
    assert int(v[2]) == int(o["converged"]) and int(v[3]) == o["iterations"]
    T = np.array([float(x) for x in v[4:20]]).reshape(4, 4).T
    assert np.abs(T[:3, :3] - o["T"][:3, :3]).max() <= 1e-4 and np.linalg.norm(T[:3, 3] - o["T"][:3, 3]) <= 1e-3
    ref.add_points(scan1, pose)
    assert int(v[20]) == len(ref)
    c = nn.astype(np.float64)
    assert abs(float(v[21]) - (c[:, 0] + 2 * c[:, 1] + 3 * c[:, 2] + c[:, 3]).sum()) <= 1e-2


def _build_c_demo(tmp_path):
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "c_abi_demo.c"), "-o", str(exe), "-L", libdir, "-licpgpu",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_plain_c_binding_compiles_as_c99_and_fails_loudly_without_gpu(built, tmp_path):
    import torch
    exe = _build_c_demo(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    src, tgt, _ = synth.make_pair(100, 100, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_plain_c_binding_matches_oracle(built, tmp_path):
    """INTEGRATION.md section 3 as a C99 program: align + fitness, promote, align the other way round."""
    exe = _build_c_demo(tmp_path)
    src, tgt, _ = synth.make_pair(6000, 7000, seed=3)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 0, r.stderr
    lines = [ln.split() for ln in r.stdout.strip().splitlines()]
    for vals, (s, t) in zip(lines, ((src, tgt), (tgt, src))):
        ref = oracle.icp_align(s, t, oracle.default_params(), want_fitness=True)
        T = np.array([float(v) for v in vals[4:20]]).reshape(4, 4).T
        assert int(vals[0]) == int(ref["converged"]) and int(vals[1]) == ref["iterations"] and int(vals[2]) == ref["n_corr"]
        assert abs(float(vals[3]) - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
        assert np.abs(T[:3, :3] - ref["T"][:3, :3]).max() <= 1e-4 and np.linalg.norm(T[:3, 3] - ref["T"][:3, 3]) <= 1e-3


# ---- the boundary exactly as integrated: fresh VoxelGrid + fresh GICP object per scan, host clouds, rotating threads ----------
def _run_pipeline(exe, tmp_path, a, b, n_scans, leaf, iters, threads, env=None):
    pa, pb = tmp_path / "a.bin", tmp_path / "b.bin"
    a.tofile(pa)
    b.tofile(pb)
    return subprocess.run([str(exe), str(pa), str(a.shape[0]), str(pb), str(b.shape[0]), str(n_scans), repr(leaf), str(iters),
                           str(threads), "2"], capture_output=True, text=True, env=dict(os.environ, **(env or {})))


def test_odometer_pipeline_demo_compiles_and_fails_loudly_without_gpu(built, tmp_path):
    import torch
    exe = _build_demo(tmp_path, "odometer_pipeline_demo")
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    a, b, _ = synth.make_pair(100, 100, seed=1)
    r = _run_pipeline(exe, tmp_path, a, b, 4, 0.2, 10, 2)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_odometer_pipeline_as_integrated_matches_the_resident_pipeline_bit_for_bit(built, tmp_path):
    """laserCloudCallback with the two swapped type names (tests/cpp/odometer_pipeline_demo.cpp), callbacks rotating over 4
    threads: every scan's transform, iteration count and fitness equal the C-ABI pipeline's (device voxel filter -> GICP ->
    fitness -> promote on ONE context) bit for bit -- with icpgpu_set_target's recognition of the previous source and
    without it (ICPGPU_RECOGNISE=0: every target uploaded and rebuilt)."""
    from icpslam_amd import GICP, Context
    exe = _build_demo(tmp_path, "odometer_pipeline_demo")
    a, b, _ = synth.make_pair(60000, 60000, seed=21)
    n_scans, leaf = 9, 0.2
    want = []
    with Context(0) as c:
        c.set_params(c.default_params(), method=GICP, max_iterations=10)
        for k in range(n_scans):
            c.set_source_voxel_filtered((a, b)[k % 2], leaf)
            if k == 0:
                c.promote_source_to_target()
                continue
            r = c.align(want_fitness=True)
            want.append(r)
            if r["converged"] and r["fitness"] < 20:
                c.promote_source_to_target()
    assert len(want) == n_scans - 1
    rates = {}
    for threads, env in ((4, {}), (1, {}), (4, {"ICPGPU_RECOGNISE": "0"})):
        r = _run_pipeline(exe, tmp_path, a, b, n_scans, leaf, 10, threads, env)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        assert lines[-1].startswith("TIMING")
        rates[(threads, bool(env))] = float(lines[-1].split()[6])
        rows = [ln.split() for ln in lines[:-1]]
        assert len(rows) == len(want)
        for row, ref in zip(rows, want):
            T = np.array([float(v) for v in row[5:21]], np.float32).reshape(4, 4).T
            assert int(row[1]) == int(ref["converged"]) and int(row[2]) == ref["iterations"]
            assert np.array_equal(T, ref["T"]), (row[0], threads, env)
            assert float(row[3]) == ref["fitness"]
    print("shim pipeline scans/s:", rates)


@pytest.mark.gpu
def test_odometer_pipeline_with_the_quadratic_inner_solver(built, tmp_path):
    """The UNCHANGED pipeline binary with ICPGPU_GICP_INNER=quadratic in its environment (include/icpgpu.h: icpgpu_gicp_inner) against
    the C-ABI pipeline with icpgpu_params.gicp_inner = QUADRATIC: bit for bit (the mode's sums are exact, the host arithmetic is
    the same), and close to the default mode's transforms on this pair (1e-4 in R, 1 cm in t: the
    registration of such pairs is only defined to millimetres, see the comment below)."""
    from icpslam_amd import GICP, GICP_INNER_EXACT, GICP_INNER_QUADRATIC, Context
    exe = _build_demo(tmp_path, "odometer_pipeline_demo")
    a, b, _ = synth.make_pair(60000, 60000, seed=21)
    n_scans, leaf = 7, 0.2
    want = {}
    for inner in (GICP_INNER_QUADRATIC, GICP_INNER_EXACT):
        want[inner] = []
        with Context(0) as c:
            c.set_params(c.default_params(), method=GICP, max_iterations=10, gicp_inner=inner)
            for k in range(n_scans):
                c.set_source_voxel_filtered((a, b)[k % 2], leaf)
                if k == 0:
                    c.promote_source_to_target()
                    continue
                r = c.align(want_fitness=True)
                want[inner].append(r)
                if r["converged"] and r["fitness"] < 20:
                    c.promote_source_to_target()
    r = _run_pipeline(exe, tmp_path, a, b, n_scans, leaf, 10, 4, {"ICPGPU_GICP_INNER": "quadratic"})
    assert r.returncode == 0, r.stderr
    rows = [ln.split() for ln in r.stdout.strip().splitlines()[:-1]]
    assert len(rows) == n_scans - 1
    for row, ref, exact in zip(rows, want[GICP_INNER_QUADRATIC], want[GICP_INNER_EXACT]):
        T = np.array([float(v) for v in row[5:21]], np.float32).reshape(4, 4).T
        assert int(row[1]) == int(ref["converged"]) and int(row[2]) == ref["iterations"]
        assert np.array_equal(T, ref["T"]), row[0]
        assert float(row[3]) == ref["fitness"]
        assert np.abs(T[:3, :3].astype(np.float64) - exact["T"][:3, :3]).max() <= 1e-4
        # (offline, scripts/r5/quadratic_costing.py: PCL's own sums run backwards move a pipeline pair by 0.55 mm in the median, 3.1 mm
        #  at worst; this mode by 0.7-1.1 mm, 3.3-7.8 mm at worst)
        assert np.linalg.norm(T[:3, 3].astype(np.float64) - exact["T"][:3, 3]) <= 1e-2


@pytest.mark.gpu
def test_gicp_shim_quadratic_setter_equals_the_parameter(built, tmp_path):
    """icp.setQuadraticInnerSolver(true) on the shim's GICP object == icpgpu_params.gicp_inner = QUADRATIC through the C-ABI: the same
    transform, iteration count and fitness, bit for bit."""
    from icpslam_amd import GICP, GICP_INNER_QUADRATIC, Context
    exe = _build_demo(tmp_path, "gicp_shim_demo")
    src, tgt, _ = synth.make_pair(6000, 6500, seed=3)
    r = _run(exe, tmp_path, src, tgt, 10, "quadratic")
    assert r.returncode == 0, r.stderr
    l1 = r.stdout.strip().splitlines()[0].split()
    with Context(0) as c:
        c.set_params(c.default_params(), method=GICP, max_iterations=10, gicp_inner=GICP_INNER_QUADRATIC)
        c.set_source(src)
        c.set_target(tgt)
        ref = c.align(want_fitness=True)
    T = np.array([float(v) for v in l1[3:19]], np.float32).reshape(4, 4).T
    assert int(l1[0]) == int(ref["converged"]) and int(l1[1]) == ref["iterations"]
    assert np.array_equal(T, ref["T"])
    assert float(l1[2]) == ref["fitness"]
