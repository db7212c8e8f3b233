"""Round 6 (icpgpu.h 1.1): the aligned cloud staged by the transform kernel itself, in front of the fitness sweep -- through
icpgpu_align (copied to the caller's buffer while the sweep runs) and icpgpu_align_view (handed out as a view of the staging
buffer).  What /root/reference/src/icpslam/icp_odometer.cpp:196-201 asks for on every scan: align(output), then getFitnessScore."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from icpslam_amd import GICP, P2P_SVD as P2P, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("method", [P2P, GICP])
def test_aligned_cloud_is_the_oracles_whichever_way_it_leaves(ctx, method):
    for n, fitness in ((3000, True), (30000, True), (30000, False), (120000, True)):
        src, tgt, _ = synth.make_pair(n, n + 100, seed=500 + n)
        ctx.set_params(ctx.default_params(), method=method, max_iterations=10)
        ctx.set_source(src)
        ctx.set_target(tgt)
        a = ctx.align(want_cloud=True, want_fitness=fitness)
        v = ctx.align_view(want_fitness=fitness)
        ref = oracle.icp_align(src, tgt, oracle.default_params(method=method, max_iterations=10), want_fitness=fitness, want_cloud=True)
        for r in (a, v):
            assert (r["iterations"], r["n_corr"], r["converged"]) == (ref["iterations"], ref["n_corr"], ref["converged"])
            assert np.array_equal(_bits(r["T"]), _bits(v["T"]))
            if fitness:
                assert r["fitness"] == a["fitness"] and abs(r["fitness"] - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
            assert r["cloud"].shape == (n, 4) and np.array_equal(_bits(r["cloud"]), _bits(a["cloud"]))
        # the cloud is T_final applied to the source with the arithmetic contract's fused operations: the
        # oracle's output cloud, bit for bit, whenever the transforms agree to the bit (GICP's do; P2P's differ by summation order)
        if np.array_equal(_bits(a["T"]), _bits(np.asarray(ref["T"], np.float32))):
            assert np.array_equal(_bits(a["cloud"]), _bits(ref["cloud"]))
        else:
            assert np.abs(a["cloud"][:, :3] - ref["cloud"][:, :3]).max() <= 1e-3


def test_million_point_source_goes_through_the_copy_engine_and_views_grow_the_staging_buffer(ctx):
    src, tgt, _ = synth.make_pair(600000, 50000, seed=71)   # 9.6 MB of aligned cloud: beyond the staging buffer's 8 MB
    ctx.set_params(ctx.default_params(), method=P2P, max_iterations=3)
    ctx.set_source(src)
    ctx.set_target(tgt)
    a = ctx.align(want_cloud=True, want_fitness=True)
    v = ctx.align_view(want_fitness=True)
    assert np.array_equal(_bits(a["T"]), _bits(v["T"])) and a["fitness"] == v["fitness"]
    assert np.array_equal(_bits(a["cloud"]), _bits(v["cloud"]))
    T = a["T"].astype(np.float32)
    want = oracle.transform_cloud(src, T) if hasattr(oracle, "transform_cloud") else None
    if want is not None:
        assert np.array_equal(_bits(a["cloud"]), _bits(want))


def test_staged_and_copy_engine_clouds_have_the_same_bits(tmp_path):
    """Development flavour, ICPGPU_STAGE_DIRECT=0: the aligned cloud goes through device memory and the copy engine behind the
    fitness sweep, as until round 5 -- the same bits as the staged way."""
    code = (
        "import sys, numpy as np\n"
        "from icpslam_amd import Context, GICP, P2P_SVD as P2P, synth\n"
        "with Context(0) as c:\n"
        "    for method in (P2P, GICP):\n"
        "        src, tgt, _ = synth.make_pair(40000, 41000, seed=9)\n"
        "        c.set_params(c.default_params(), method=method, max_iterations=10)\n"
        "        c.set_source(src); c.set_target(tgt)\n"
        "        r = c.align(want_cloud=True, want_fitness=True)\n"
        "        print(r['T'].tobytes().hex()[:64], float(r['fitness']).hex(), __import__('hashlib').sha256(r['cloud'].tobytes()).hexdigest())\n")
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, ICPGPU_FLAVOUR="dev", ICPGPU_STAGE_DIRECT=flag, PYTHONPATH=ROOT)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.strip().splitlines())
    assert outs[0] == outs[1] and len(outs[0]) == 2
