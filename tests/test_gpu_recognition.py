"""icpgpu_set_target recognises a cloud the context already holds (include/icpgpu.h): the reference hands scan k-1's source
back as scan k's target (`*prev_cloud_ = *curr_cloud_`, /root/reference/src/icpslam/icp_odometer.cpp:209 then :194)."""
import numpy as np
import pytest

from icpslam_amd import GICP, Context, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["p2p", "gicp"])
@pytest.mark.parametrize("n", [30000, 120000])
def test_previous_source_is_recognised_and_results_do_not_change(built, method, n):
    a, b, _ = synth.make_pair(n, n, seed=31)
    c3, _, _ = synth.make_pair(n, n, seed=32)
    kw = dict(max_iterations=6)
    if method == "gicp":
        kw["method"] = GICP
    with Context(0) as fresh:                      # what a context without history answers for (source c3, target a)
        fresh.set_params(fresh.default_params(), **kw)
        fresh.set_source(c3)
        fresh.set_target(a)
        want = fresh.align(want_fitness=True)
    with Context(0) as c:
        c.set_params(c.default_params(), **kw)
        c.set_target(b)
        c.set_source(a)
        c.align()
        before = c.profile()
        c.set_target(a.copy())                     # the previous source, in another host buffer: the promote path
        mid = c.profile()
        assert mid.targets_recognised == before.targets_recognised + 1
        assert c.n_target == n and c.n_source == n   # (the source stays set: a device-side copy of the cloud)
        c.set_source(c3)
        got = c.align(want_fitness=True)
        assert np.array_equal(got["T"], want["T"]) and got["iterations"] == want["iterations"]
        # (fitness: a float64 sum over the source in CELL order from 100k points on, and the order inside a cell comes from
        # atomic ranks -- the last bit may differ from run to run of the SAME call sequence, with or without recognition)
        assert got["n_corr"] == want["n_corr"] and abs(got["fitness"] - want["fitness"]) <= 1e-12 * want["fitness"]
        c.set_target(a.copy())                     # the unchanged target (a rejected scan keeps prev_cloud_): nothing to do
        after = c.profile()
        assert after.targets_recognised == mid.targets_recognised + 1 and after.grid_builds == c.profile().grid_builds
        again = c.align(want_fitness=True)
        assert np.array_equal(again["T"], want["T"]) and abs(again["fitness"] - want["fitness"]) <= 1e-12 * want["fitness"]
        # same size, one bit different: NOT recognised, uploaded -- in a point the 256-point sample of the quick look holds
        # (n/2) and, below, in one only the full fingerprint sees
        a2 = a.copy()
        a2.view(np.uint32)[n // 2, 1] ^= 1
        c.set_target(a2)
        assert c.profile().targets_recognised == after.targets_recognised
        a3 = a2.copy()
        a3.view(np.uint32)[n // 2 + 1, 2] ^= 1
        c.set_target(a3)
        assert c.profile().targets_recognised == after.targets_recognised
        c.set_target(a3.copy())
        assert c.profile().targets_recognised == after.targets_recognised + 1
        c.set_target(a2)
        with Context(0) as fresh2:
            fresh2.set_params(fresh2.default_params(), **kw)
            fresh2.set_source(c3)
            fresh2.set_target(a2)
            want2 = fresh2.align()
        got2 = c.align()
        assert np.array_equal(got2["T"], want2["T"]) and got2["n_corr"] == want2["n_corr"]


def test_device_and_host_fingerprints_agree(built):
    """The recognition compares a host hash of the new buffer with a device hash of the resident cloud: sizes around the
    kernel's block and grid boundaries, NaNs and negative zeros included."""
    rng = np.random.default_rng(5)
    with Context(0) as c:
        for n in (1, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097, 300001, 1048577 + 5):
            a = rng.normal(size=(n, 4)).astype(np.float32)
            a[0, :] = (-0.0, np.nan, np.inf, 1.0)
            c.set_source(a)
            base = c.profile().targets_recognised
            c.set_target(a.copy())
            assert c.profile().targets_recognised == base + 1, n


def test_a_cloud_registered_against_itself_survives_the_recognition(built):
    """set_source(A); set_target(A): the target is recognised as the current source -- and the source must still be there."""
    a, _, _ = synth.make_pair(40000, 40000, seed=41)
    with Context(0) as c:
        c.set_params(c.default_params(), max_iterations=3)
        c.set_source(a)
        c.set_target(a.copy())
        assert c.profile().targets_recognised == 1
        r = c.align(want_fitness=True)
        assert r["converged"] and r["n_corr"] == 40000 and r["fitness"] <= 1e-10
        assert np.abs(r["T"] - np.eye(4, dtype=np.float32)).max() <= 1e-6
        idx, d2 = c.nn(np.eye(4))
        assert np.array_equal(idx, np.arange(40000)) and not d2.any()


@pytest.mark.parametrize("method", ["p2p", "gicp"])
def test_the_voxel_filters_result_is_adopted_as_the_source(built, method):
    """voxelFilterCloud -> setInputSource (/root/reference/src/icpslam/icp_odometer.cpp:177,193): the filtered cloud the caller
    hands back is still in HBM -- icpgpu_set_source adopts it (no upload, the raw scan's box for the grid builds) and nothing
    about the results changes; a cloud that differs in one bit, or has another size, is uploaded as before."""
    raw_a, raw_b, _ = synth.make_pair(120000, 120000, seed=51)
    kw = dict(max_iterations=6)
    if method == "gicp":
        kw["method"] = GICP
    with Context(0) as fresh:                      # upload path only: filter in one context, register in another
        fa, fb = fresh.voxel_grid(raw_a, 0.2), fresh.voxel_grid(raw_b, 0.2)
    with Context(0) as plain:
        plain.set_params(plain.default_params(), **kw)
        plain.set_target(fb)
        plain.set_source(fa)
        assert plain.profile().sources_adopted == 0
        want = plain.align(want_fitness=True)
    with Context(0) as c:
        c.set_params(c.default_params(), **kw)
        gb = c.voxel_grid(raw_b, 0.2)
        assert np.array_equal(gb, fb)
        c.set_target(gb)
        ga = c.voxel_grid(raw_a, 0.2)
        c.set_source(ga.copy())                    # another host buffer, the same bytes
        assert c.profile().sources_adopted == 1 and c.n_source == ga.shape[0]
        got = c.align(want_fitness=True)
        assert np.array_equal(got["T"], want["T"]) and got["iterations"] == want["iterations"]
        assert got["n_corr"] == want["n_corr"] and abs(got["fitness"] - want["fitness"]) <= 1e-12 * want["fitness"]
        assert np.array_equal(c.voxel_grid(raw_a, 0.2), ga)          # the filter's own buffer is untouched by the adoption
        # the next scan: the previous source comes back as the target (promote path), the new filtered scan is adopted again
        c.set_target(ga.copy())
        gb2 = c.voxel_grid(raw_b, 0.2)
        c.set_source(gb2)
        p = c.profile()
        assert p.sources_adopted == 2 and p.targets_recognised >= 1
        back = c.align(want_fitness=True)
        with Context(0) as plain2:
            plain2.set_params(plain2.default_params(), **kw)
            plain2.set_target(fa)
            plain2.set_source(fb)
            want_back = plain2.align(want_fitness=True)
        assert np.array_equal(back["T"], want_back["T"]) and back["n_corr"] == want_back["n_corr"]
        # one bit off (in a sampled point, then in one only the full fingerprint sees), or one point short: uploaded
        ga = c.voxel_grid(raw_a, 0.2)
        m = ga.shape[0]
        x = ga.copy(); x.view(np.uint32)[m // 2, 0] ^= 1
        c.set_source(x)
        y = ga.copy(); y.view(np.uint32)[m // 2 + 1, 2] ^= 1
        c.set_source(y)
        c.set_source(ga[:-1])
        assert c.profile().sources_adopted == 2
        idx, d2 = c.nn(np.eye(4))                  # ... and what was uploaded is what is searched
        assert idx.shape[0] == m - 1
