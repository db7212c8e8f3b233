"""DEVELOPMENT FLAVOUR (libicpgpu_dev.so; icp_tile.hip is not in the release library).
The experimental grid search on the matrix cores (icp_tile.hip, ICPGPU_TILE_SEARCH=1; the correspondence search PCL runs per
ICP iteration, /root/reference/src/icpslam/icp_odometer.cpp:198): whole alignments equal the shipped grid search's bit for
bit -- transform, correspondences, iterations, fitness -- and the oracle's within the contract's tolerance."""
import os

import numpy as np
import pytest

import oracle
from icpslam_amd import Context, synth

pytestmark = pytest.mark.gpu


def _run(ctx, src, tgt, **kw):
    ctx.set_params(ctx.default_params(), **kw)
    ctx.set_source(src); ctx.set_target(tgt)
    return ctx.align(want_fitness=True)


@pytest.mark.parametrize("case", ["lidar 30k", "lidar 60k forced", "outliers and non-finite points", "threshold 0.3 m", "scan vs submap"])
def test_alignments_equal_the_shipped_grid_search(built, dev_flavour, case):
    if dev_flavour.delegated:
        return
    kw = dict(max_iterations=12)
    if case == "lidar 30k":
        src, tgt, _ = synth.make_pair(30000, 28000, seed=21)
    elif case == "lidar 60k forced":
        src, tgt, _ = synth.make_pair(60000, 60000, seed=22)
        kw = dict(max_iterations=8, force_iterations=1)
    elif case == "outliers and non-finite points":
        src, tgt, _ = synth.make_pair(20000, 24000, seed=23)
        src = src.copy(); tgt = tgt.copy()
        src[100:140, :3] += (300.0, -200.0, 40.0)            # far outliers inside Morton groups: huge search boxes
        src[7, :3] = np.nan
        tgt[9, :3] = np.inf
        tgt[500:520] = tgt[500]                              # duplicates: ties, lowest index wins
    elif case == "threshold 0.3 m":
        src, tgt, _ = synth.make_pair(40000, 40000, seed=24)
        kw = dict(max_iterations=10, max_correspondence_distance=0.3)
    else:
        src, tgt, _ = synth.make_scan_vs_submap(20000, 150000, seed=25)
    with Context(0) as c:
        want = _run(c, src, tgt, **kw)
        os.environ["ICPGPU_TILE_SEARCH"] = "1"
        try:
            got = _run(c, src, tgt, **kw)
        finally:
            del os.environ["ICPGPU_TILE_SEARCH"]
    assert np.array_equal(got["T"], want["T"]), case
    assert (got["n_corr"], got["iterations"], got["converged"]) == (want["n_corr"], want["iterations"], want["converged"])
    assert got["fitness"] == want["fitness"] or abs(got["fitness"] - want["fitness"]) <= 1e-12 * abs(want["fitness"])
    if case == "lidar 30k":
        o = oracle.icp_align(src, tgt, oracle.default_params(**kw))
        assert got["iterations"] == o["iterations"] and got["n_corr"] == o["n_corr"]
        assert np.abs(got["T"][:3, :3] - o["T"][:3, :3]).max() <= 1e-4   # BASELINE tolerance (R)
        assert np.linalg.norm(got["T"][:3, 3] - o["T"][:3, 3]) <= 1e-3   # BASELINE tolerance (t), metres
