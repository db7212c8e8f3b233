"""The brute-force search with its lower bound on the bf16 matrix path (icp_brute_bf16.hip; row a2 of SURVEY.md 8(a), the
nearest-neighbour arithmetic PCL reaches from /root/reference/src/icpslam/icp_odometer.cpp:198): the same keys as the plain
vector kernel and the f32-MFMA kernel bit for bit, and -- in the kernel's test mode, every pair evaluated exactly -- not one
pair whose bound exceeds what its own distance allows."""
import os

import numpy as np
import pytest

import oracle
from icpslam_amd import NN_BRUTE, Context, synth

pytestmark = pytest.mark.gpu

TAU = 2.0 ** -13


def _cloud(p):
    c = np.ones((p.shape[0], 4), np.float32)
    c[:, :3] = p
    return c


def _cases():
    rng = np.random.default_rng(77)
    a, b, _ = synth.make_pair(30000, 26000, seed=5)
    yield "lidar pair", a, b, synth.pose_matrix(0.3, -0.2, 0.05, 0.01, -0.02, 0.04)
    s, m, _ = synth.make_scan_vs_submap(20000, 90000, seed=6)
    yield "scan vs submap", s, m, synth.pose_matrix(-0.4, 0.3, 0.0, 0.0, 0.01, -0.03)
    far = a.copy(); far[:, :3] += (1.0e4, -2.0e4, 300.0)
    farb = b.copy(); farb[:, :3] += (1.0e4, -2.0e4, 300.0)
    yield "coordinates of 1e4 m", far, farb, np.eye(4)
    yield "millimetre scene", _cloud(a[:, :3] * 1e-3), _cloud(b[:, :3] * 1e-3), np.eye(4)
    yield "scene of 1e-9 m (below the scale the bound is claimed for)", _cloud(a[:9000, :3] * 1e-10), _cloud(b[:9000, :3] * 1e-10), np.eye(4)
    c = rng.uniform(-20, 20, (12, 3))
    clumps = _cloud((c[rng.integers(0, 12, 24000)] + rng.normal(0, 0.05, (24000, 3))).astype(np.float32))
    clumps[100:400] = clumps[100]                                     # exact duplicates: ties, lowest index wins
    q = clumps[rng.permutation(24000)[:16000]].copy()
    q[:50, :3] = (900.0, -700.0, 80.0)                                # far outliers inside a workgroup: a large P
    q[60, :3] = np.nan
    t = clumps.copy()
    t[7, :3] = np.inf
    t[8, :3] = np.nan
    yield "clumps, duplicates, outliers, non-finite points", q, t, np.eye(4)
    g = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(10)), -1).reshape(-1, 3).astype(np.float32) * 0.25
    yield "lattice: every distance tied many times", _cloud(g + 0.125), _cloud(g), np.eye(4)


@pytest.mark.parametrize("name,src,tgt,T", list(_cases()), ids=[c[0] for c in _cases()])
def test_three_brute_kernels_agree_and_no_bound_is_too_high(built, dev_flavour, name, src, tgt, T):
    if dev_flavour.delegated:     # ICPGPU_MFMA_CHECK_BOUND exists only in the development flavour (icp_env.h)
        return
    T = np.asarray(T, np.float32)
    os.environ["ICPGPU_MFMA_CHECK_BOUND"] = "1"
    try:
        with Context(0) as c:
            got = {}
            for variant in (0, 2, 1):
                c.set_params(c.default_params(), nn_mode=NN_BRUTE, brute_variant=variant)
                c.set_source(src); c.set_target(tgt)
                c.profile_reset()
                got[variant] = c.nn(T)
                if variant == 0:
                    p = c.profile()
                    assert p.brute_bound_violations == 0, name
                    assert p.brute_bound_worst < TAU / 2, (name, p.brute_bound_worst)   # (the budget in the kernel's header: half of tau)
                    worst = p.brute_bound_worst
    finally:
        del os.environ["ICPGPU_MFMA_CHECK_BOUND"]
    for variant in (2, 1):
        assert np.array_equal(got[0][0], got[variant][0]), (name, variant)
        assert np.array_equal(got[0][1].view(np.uint32), got[variant][1].view(np.uint32)), (name, variant)
    idx, d2 = oracle.nn(src, tgt, T)
    assert np.array_equal(got[0][0], idx) and np.array_equal(got[0][1].view(np.uint32), d2.view(np.uint32))
    print(f"{name}: worst excess {worst:.3e} of (P^2 + |v|^2), tau = {TAU:.3e}")


def test_seeded_sweeps_and_the_production_kernel_without_the_check(built):
    """The kernel as it ships (pipelined, no test mode): a forced 4-iteration alignment in brute-force mode -- the first sweep
    unseeded, the next ones seeded with the previous neighbours -- equals the plain vector kernel's, transform bits included."""
    src, tgt, _ = synth.make_pair(40000, 36000, seed=8)
    res = {}
    with Context(0) as c:
        for variant in (0, 1):
            c.set_params(c.default_params(), nn_mode=NN_BRUTE, brute_variant=variant, max_iterations=4, force_iterations=1)
            c.set_source(src); c.set_target(tgt)
            res[variant] = c.align(want_fitness=True)
        assert c.profile().brute_bound_violations == 0
    assert np.array_equal(res[0]["T"], res[1]["T"]) and res[0]["n_corr"] == res[1]["n_corr"]
    assert res[0]["fitness"] == res[1]["fitness"]


def test_bound_campaign_slice_of_200_random_clouds(built):
    """A 200-cloud slice of scripts/bf16_campaign.py (4000 clouds in round 3) with the kernel's test mode on -- EVERY pair's
    bound against its exact distance: shapes, scales 1e-3 .. 1e3, offsets to 1e4 m, lattices, duplicates, outliers, non-finite
    points.  Development flavour (ICPGPU_MFMA_CHECK_BOUND exists only there).  The MFMA's accumulation error inside the
    budget is measured separately (scripts/probes/mfma_accum.cpp, profiles/r04_mfma_accumulation.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ICPGPU_FLAVOUR="dev", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "bf16_campaign.py"), "4000", "4200"], env=env, capture_output=True,
                         text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert "0 clouds with different keys, 0 bound violations" in last, last
    worst = float(last.split("worst excess")[1].split()[0])
    assert 0.0 < worst < TAU / 2, last
    print(last)
