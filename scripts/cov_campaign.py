"""Dev tool (round 6): GICP's covariances (gicp_cov_select_kernel -> gicp_cov_far_kernel -> gicp_cov_kernel -> finish) against the oracle on
many random clouds -- sizes 20..60k, gaussian / uniform / raw-scan / voxel-filtered-scan / clustered shapes, duplicates, lattices and
non-finite points.  Usage: python scripts/cov_campaign.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from icpslam_amd import Context, GICP, synth

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
refused = 0
worst = 0.0
inexact = 0
points = 0
t0 = time.time()
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), method=GICP)
    scene = synth.make_scene(3)
    for seed in range(first, last):
        rng = np.random.default_rng(70_000 + seed)
        n = int(rng.integers(20, 60000))
        kind = seed % 6
        c = np.ones((n, 4), np.float32)
        if kind == 0:
            c[:, :3] = rng.normal(0, float(rng.choice([0.5, 5.0, 60.0])), (n, 3)).astype(np.float32)
        elif kind == 1:
            c[:, :3] = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
        elif kind == 2:
            c = synth.scan(scene, np.eye(4), n, seed=seed)
        elif kind == 3:
            c = oracle.voxel_grid(synth.scan(scene, synth.pose_matrix(float(rng.uniform(-20, 20)), 0, 0, 0, 0, 0), 4 * n, seed=seed), float(rng.choice([0.1, 0.2, 0.4])))
        elif kind == 4:  # a few tight clusters in a sparse volume, far-away stragglers
            k = int(rng.integers(1, 6))
            centres = rng.uniform(-50, 50, (k, 3))
            c[:, :3] = (centres[rng.integers(0, k, n)] + rng.normal(0, 0.3, (n, 3))).astype(np.float32)
            c[::11, :3] = rng.uniform(-200, 200, (len(c[::11]), 3)).astype(np.float32)
        else:  # a lattice with duplicates: dozens of equal distances
            m = max(3, int(round(n ** (1 / 3))))
            g = (np.arange(m, dtype=np.float32) * np.float32(rng.choice([0.1, 0.25, 1.0])))
            c = np.ones((m ** 3, 4), np.float32)
            c[:, :3] = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
            c = np.concatenate([c, c[: len(c) // 7]])
        if len(c) > 40 and seed % 5 == 0:
            c[rng.integers(0, len(c), 3), rng.integers(0, 3, 3)] = np.nan
        fin = np.isfinite(c[:, :3]).all(axis=1)
        if fin.sum() < 20:
            continue
        ctx.set_source(c)
        try:
            got = ctx.gicp_covariances()[fin]
        except Exception as e:  # (a cloud the k-NN grid refuses -- thousands of points in one cell -- is an error, not a wrong answer)
            refused += 1
            print(f"refused seed {seed} kind {kind} n {len(c)}: {str(e)[:100]}", flush=True)
            continue
        ref = oracle.gicp_covariances(c[fin])
        diff = np.abs(got - ref).reshape(len(ref), -1).max(axis=1)
        points += len(ref)
        inexact += int((diff > 0).sum())
        worst = max(worst, float(diff.max()))
        if diff.max() > 1e-6 or (diff > 0).mean() > 0.002:
            bad += 1
            print(f"MISMATCH seed {seed} kind {kind} n {len(c)}: {int((diff > 0).sum())} points differ, worst {diff.max():.3e}", flush=True)
print(f"covariance clouds {first}..{last}: {bad} mismatches, {refused} refused; {inexact} of {points} points not bit-identical to the oracle, worst difference {worst:.3e}; {time.time()-t0:.0f} s")
