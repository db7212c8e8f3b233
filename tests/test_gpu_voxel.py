"""GPU parity of the voxel-grid filter against the oracle's restatement of pcl::VoxelGrid."""
import numpy as np
import pytest

import oracle
from icpslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,leaf,seed", [(1, 0.2, 0), (1000, 0.2, 1), (20000, 0.2, 2), (50000, 0.05, 3), (50000, 1.0, 4),
                                           (200000, 0.2, 5)])
def test_voxel_grid_bit_exact(ctx, n, leaf, seed):
    cloud, _, _ = synth.make_pair(n, 10, seed=seed)
    got = ctx.voxel_grid(cloud, leaf)
    ref = oracle.voxel_grid(cloud, leaf)
    assert got.shape == ref.shape                      # number of occupied cells and their order: exact
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))   # float32 means accumulated in the same order


def test_voxel_grid_edge_cases(ctx):
    e = np.zeros((0, 4), np.float32)
    assert ctx.voxel_grid(e, 0.2).shape == (0, 4)
    cloud, _, _ = synth.make_pair(3000, 10, seed=7)
    bad = cloud.copy()
    bad[5, 0] = np.nan
    bad[9, 2] = np.inf
    got = ctx.voxel_grid(bad, 0.2)
    ref = oracle.voxel_grid(np.delete(cloud, [5, 9], axis=0), 0.2)    # non-finite points are skipped
    assert np.array_equal(got, ref)
    # leaf so small that the cell index space overflows int32: PCL returns the input unchanged
    tiny = ctx.voxel_grid(cloud, 1e-4)
    assert np.array_equal(tiny, cloud)
    assert np.array_equal(oracle.voxel_grid(cloud, 1e-4), cloud)
    # all points in one cell
    one = np.ones((500, 4), np.float32)
    one[:, :3] = np.random.default_rng(0).uniform(0.01, 0.19, (500, 3))
    got = ctx.voxel_grid(one, 0.2)
    assert got.shape == (1, 4) and np.array_equal(got, oracle.voxel_grid(one, 0.2))
    with pytest.raises(Exception):
        ctx.voxel_grid(cloud, 0.0)


def test_filtered_source_feeds_icp(ctx):
    """The odometer's sequence: voxelFilterCloud(input -> curr) then ICP(curr, prev) (icp_odometer.cpp:177-198)."""
    a, b, _ = synth.make_pair(40000, 40000, seed=9)
    fa, fb = oracle.voxel_grid(a, 0.2), oracle.voxel_grid(b, 0.2)
    ctx.set_params(ctx.default_params())
    n = ctx.set_source_voxel_filtered(a, 0.2)
    assert n == fa.shape[0]
    ctx.set_target(ctx.voxel_grid(b, 0.2))
    got = ctx.align(want_fitness=True)
    ref = oracle.icp_align(fa, fb, want_fitness=True)
    assert (got["iterations"], got["n_corr"], got["converged"]) == (ref["iterations"], ref["n_corr"], ref["converged"])
    assert np.abs(got["T"] - ref["T"]).max() <= 1e-4
    assert abs(got["fitness"] - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_voxel_filter_against_oracle(ctx, seed):
    import oracle
    rng = np.random.default_rng(700 + seed)
    n = int(rng.integers(1, 60000))
    leaf = float(rng.choice([0.03, 0.1, 0.2, 0.77, 5.0]))
    scale = float(rng.choice([2.0, 30.0, 300.0]))
    cloud = np.ones((n, 4), np.float32)
    cloud[:, :3] = rng.normal(0, scale, (n, 3)).astype(np.float32)
    if n > 40:
        cloud[10:30] = cloud[10]                                  # exact duplicates share a voxel
    out = ctx.voxel_grid(cloud, leaf)
    ref = oracle.voxel_grid(cloud, leaf)                          # returns the input unchanged on index overflow, like PCL
    assert out.shape == ref.shape and np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("leaf", [0.2, 0.5, 2.0])
def test_raw_scan_dense_voxels_bit_exact(ctx, leaf):
    """A raw scan's near-field voxels hold hundreds of points: the direct path's in-LDS sort must keep PCL's input order
    inside every voxel (float sums), and groups of several buckets / one bucket filling a group must both work."""
    scene = synth.make_scene(31)
    for n in (5000, 70000, 200000):
        cloud = synth.scan(scene, np.eye(4), n, seed=n)
        got, ref = ctx.voxel_grid(cloud, leaf), oracle.voxel_grid(cloud, leaf)
        assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref)), (n, leaf)


def test_voxel_over_capacity_falls_back_to_the_sort_path(ctx):
    """Thousands of points in ONE voxel exceed what a workgroup sorts in LDS: the direct path reports it and the library-sort
    path produces the result -- same bits as the oracle either way."""
    rng = np.random.default_rng(5)
    cloud = np.ones((30000, 4), np.float32)
    cloud[:, :3] = rng.uniform(-20, 20, (30000, 3)).astype(np.float32)
    cloud[1000:9000, :3] = rng.uniform(0.01, 0.19, (8000, 3)).astype(np.float32)   # 8000 points in the voxel at the origin
    got, ref = ctx.voxel_grid(cloud, 0.2), oracle.voxel_grid(cloud, 0.2)
    assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref))
    # around the capacity of one group (3840 elements) and of one bucket (3840 - 512), and the sort's padding (4096)
    for m in (3327, 3328, 3329, 3839, 3840, 3841, 4095, 4096, 4097):
        c = np.ones((m + 500, 4), np.float32)
        c[:m, :3] = rng.uniform(0.01, 0.19, (m, 3)).astype(np.float32)
        c[m:, :3] = rng.uniform(-5, 5, (500, 3)).astype(np.float32)
        got, ref = ctx.voxel_grid(c, 0.2), oracle.voxel_grid(c, 0.2)
        assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref)), m
    # and a later ordinary cloud on the same context is unaffected (the self-cleaning histogram stayed clean)
    cloud2, _, _ = synth.make_pair(20000, 10, seed=3)
    assert np.array_equal(_bits(ctx.voxel_grid(cloud2, 0.2)), _bits(oracle.voxel_grid(cloud2, 0.2)))


def test_voxel_sizes_around_the_group_quantum(ctx):
    rng = np.random.default_rng(9)
    for n in (2, 63, 64, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4097, 8191):
        c = np.ones((n, 4), np.float32)
        c[:, :3] = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
        for leaf in (0.1, 1.5):
            got, ref = ctx.voxel_grid(c, leaf), oracle.voxel_grid(c, leaf)
            assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref)), (n, leaf)


def test_voxel_narrow_and_wide_sort_words(ctx):
    """Groups of few buckets sort 32-bit words (cell key relative to the group above the point index), the others 64-bit
    ones: both against the oracle, on clouds that force each -- a tight cluster (narrow everywhere), the same cluster inside a
    huge sparse volume (wide groups next to narrow ones), and duplicates of one point (equal keys, order by index only)."""
    rng = np.random.default_rng(21)
    tight = np.ones((60000, 4), np.float32)
    tight[:, :3] = rng.normal(0, 1.5, (60000, 3)).astype(np.float32)
    mixed = tight.copy()
    mixed[::7, :3] = rng.uniform(-400, 400, (len(mixed[::7]), 3)).astype(np.float32)
    dup = np.ones((3000, 4), np.float32)
    dup[:, :3] = np.float32(0.123)
    dup[1500:, :3] = rng.uniform(-3, 3, (1500, 3)).astype(np.float32)
    for cloud in (tight, mixed, dup):
        for leaf in (0.2, 0.35):
            got, ref = ctx.voxel_grid(cloud, leaf), oracle.voxel_grid(cloud, leaf)
            assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref))


def test_voxel_filter_is_repeatable_across_calls_and_sizes(ctx):
    """The direct path keeps state between calls (a self-cleaning histogram, an epoch for the published group counts): many
    calls of changing size on one context, each against the oracle."""
    rng = np.random.default_rng(33)
    scene = synth.make_scene(5)
    for rep in range(12):
        n = int(rng.integers(100, 90000))
        cloud = synth.scan(scene, np.eye(4), n, seed=rep) if rep % 2 else synth.make_pair(n, 10, seed=rep)[0]
        leaf = float(rng.choice([0.1, 0.2, 0.4]))
        got, ref = ctx.voxel_grid(cloud, leaf), oracle.voxel_grid(cloud, leaf)
        assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref)), (rep, n, leaf)


def test_voxel_index_wraps_like_pcl(ctx):
    """PCL tests the FLOAT extents for overflow but indexes with the integer ones, which can be a cell wider: the topmost
    cells' int32 index then wraps negative and those voxels come first.  Found by scripts/voxel_campaign.py (seed 2864); the
    oracle and the sort path reproduce it, the direct path hands such clouds over."""
    rng = np.random.default_rng(9000 + 2864)
    n = int(rng.integers(1, 120000))
    leaf = float(rng.choice([0.03, 0.1, 0.2, 0.35, 0.77, 2.0, 5.0]))
    c = np.ones((n, 4), np.float32)
    c[:, :3] = rng.normal(0, float(rng.choice([2.0, 30.0, 300.0])), (n, 3)).astype(np.float32)
    got, ref = ctx.voxel_grid(c, leaf), oracle.voxel_grid(c, leaf)
    assert 0 < len(ref) < n                       # filtered, not passed through
    assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref))
    assert ref[0, 2] > ref[1, 2] + 100            # the wrapped voxel: highest z, first in the output


def test_voxel_published_count_timeout_hands_over_to_the_sort_path(tmp_path):
    """The groups of the direct path wait for their predecessors' published counts; a count that never comes (here: group 0
    answers 30 ms late, a test hook) must end in the sort path's result, not in a hang or a wrong cloud."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, oracle\n"
        "from icpslam_amd import Context, synth\n"
        "c = synth.scan(synth.make_scene(2), np.eye(4), 30000, seed=1)\n"
        "with Context(0) as ctx:\n"
        "    for leaf in (0.2, 0.4):\n"
        "        got, ref = ctx.voxel_grid(c, leaf), oracle.voxel_grid(c, leaf)\n"
        "        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))\n"
        "print('ok')\n")
    env = dict(os.environ, ICPGPU_VOXEL_TEST_STALL="1", ICPGPU_FLAVOUR="dev", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_voxel_grid_view_equals_the_oracle_and_is_recognised_as_the_source(ctx):
    """Round 6 (icpgpu.h 1.1): icpgpu_voxel_grid_view hands the filtered cloud out of the pinned staging buffer -- written there by
    a kernel behind the filter's last one, its fingerprint riding on the cell count's read-back.  Same bits as the oracle (and as
    icpgpu_voxel_grid); icpgpu_set_source recognises the host copy (no upload: sources_adopted); clouds the direct path hands to the
    sort path, PCL's pass-through case, an empty cloud and a cloud beyond the staging buffer come through the copy engine with
    the same bits."""
    scene = synth.make_scene(7)
    rng = np.random.default_rng(77)
    ctx.profile_reset()
    for rep, n in enumerate((200000, 50000, 777, 120000)):
        cloud = synth.scan(scene, np.eye(4), n, seed=60 + rep)
        got, ref = ctx.voxel_grid_view(cloud, 0.2), oracle.voxel_grid(cloud, 0.2)
        assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref)), n
        assert np.array_equal(_bits(got), _bits(ctx.voxel_grid(cloud, 0.2)))
        got = ctx.voxel_grid_view(cloud, 0.2)
        before = ctx.profile().sources_adopted
        ctx.set_source(got)
        assert ctx.profile().sources_adopted == before + 1, n
    p = ctx.profile()
    assert p.voxel_views_direct == 8, p.voxel_views_direct
    # thousands of points in one voxel: the direct path reports the cloud, the sort path runs, the copy engine brings the result
    dup = np.ones((30000, 4), np.float32)
    dup[:, :3] = np.float32(0.123)
    dup[15000:, :3] = rng.uniform(-3, 3, (15000, 3)).astype(np.float32)
    got, ref = ctx.voxel_grid_view(dup, 0.2), oracle.voxel_grid(dup, 0.2)
    assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref))
    # PCL's "leaf size is too small": the input comes back unchanged
    wide = synth.make_pair(3000, 10, seed=7)[0]
    got, ref = ctx.voxel_grid_view(wide, 1e-4), oracle.voxel_grid(wide, 1e-4)
    assert got.shape == ref.shape == wide.shape and np.array_equal(_bits(got), _bits(wide)) and np.array_equal(_bits(ref), _bits(wide))
    assert ctx.voxel_grid_view(np.empty((0, 4), np.float32), 0.2).shape == (0, 4)
    # 700k points = 11 MB: beyond the staging buffer's 8 MB, the view still holds the whole result
    big = synth.scan(scene, np.eye(4), 700000, seed=99)
    got, ref = ctx.voxel_grid_view(big, 0.05), oracle.voxel_grid(big, 0.05)
    assert got.shape == ref.shape and np.array_equal(_bits(got), _bits(ref))
    assert ctx.profile().voxel_views_direct == 8


def test_filter_queued_behind_the_box_equals_the_filter_that_waits_for_it(tmp_path):
    """Round 6: the direct path's kernels are queued behind the bounding-box pass and take their parameters from a plan the DEVICE
    derives from the box (one wait per filter instead of two).  Against the oracle on the clouds that decide the plan -- a scan, no
    finite point at all, PCL's pass-through, an index that wraps (the sort path's), a single point -- and, in the development flavour
    with ICPGPU_VOXEL_PLANNED=0 (the box first, as until round 5), the same bytes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(9000 + 2864)
    n = int(rng.integers(1, 120000))
    leaf_w = float(rng.choice([0.03, 0.1, 0.2, 0.35, 0.77, 2.0, 5.0]))
    wrap = np.ones((n, 4), np.float32)
    wrap[:, :3] = rng.normal(0, float(rng.choice([2.0, 30.0, 300.0])), (n, 3)).astype(np.float32)
    scan = synth.scan(synth.make_scene(3), np.eye(4), 150000, seed=4)
    nothing = np.full((500, 4), np.nan, np.float32)
    one = np.array([[1.5, -2.5, 0.25, 1.0]], np.float32)
    cases = dict(scan=(scan, 0.2), nothing=(nothing, 0.2), tiny_leaf=(scan[:3000], 1e-4), wrap=(wrap, leaf_w), one=(one, 0.2))
    np.savez(tmp_path / "c.npz", **{k: v[0] for k, v in cases.items()})
    code = (
        "import sys, hashlib, numpy as np\n"
        "from icpslam_amd import Context\n"
        "z = np.load(sys.argv[1]); leaves = dict(" + ", ".join(f"{k}={v[1]!r}" for k, v in cases.items()) + ")\n"
        "with Context(0) as c:\n"
        "    for k in sorted(z.files):\n"
        "        for fn in (c.voxel_grid, c.voxel_grid_view):\n"
        "            out = fn(z[k], leaves[k])\n"
        "            print(k, out.shape[0], hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest())\n")
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, ICPGPU_FLAVOUR="dev", ICPGPU_VOXEL_PLANNED=flag, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path / "c.npz")], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines())
    assert outs[0] == outs[1] and len(outs[0]) == 2 * len(cases)
    import hashlib
    want = {}
    for k, (cloud, leaf) in cases.items():
        fin = cloud[np.isfinite(cloud[:, :3]).all(axis=1)]
        ref = oracle.voxel_grid(fin, leaf) if len(fin) else np.empty((0, 4), np.float32)
        want[k] = (ref.shape[0], hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest())
    for line in outs[0]:
        k, m, h = line.split()
        assert (int(m), h) == want[k], k
