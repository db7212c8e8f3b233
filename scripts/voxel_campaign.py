"""Dev tool: the voxel filter against the oracle on many random clouds -- sizes 1..120k, leaves 0.03..5 m, gaussian / uniform /
raw-scan / clustered shapes, duplicates and non-finite points -- bit for bit (count, order, float bits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from icpslam_amd import Context, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
t0 = time.time()
with Context(0) as ctx:
    scene = synth.make_scene(3)
    for seed in range(int(os.environ.get('FIRST', '0')), N):
        rng = np.random.default_rng(9000 + seed)
        n = int(rng.integers(1, 120000))
        leaf = float(rng.choice([0.03, 0.1, 0.2, 0.35, 0.77, 2.0, 5.0]))
        kind = seed % 4
        c = np.ones((n, 4), np.float32)
        if kind == 0:
            c[:, :3] = rng.normal(0, float(rng.choice([2.0, 30.0, 300.0])), (n, 3)).astype(np.float32)
        elif kind == 1:
            c[:, :3] = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
        elif kind == 2:
            c = synth.scan(scene, np.eye(4), n, seed=seed)
        else:  # a few tight clusters in a sparse volume
            k = int(rng.integers(1, 6))
            centres = rng.uniform(-50, 50, (k, 3))
            c[:, :3] = (centres[rng.integers(0, k, n)] + rng.normal(0, 0.3, (n, 3))).astype(np.float32)
            c[::11, :3] = rng.uniform(-200, 200, (len(c[::11]), 3)).astype(np.float32)
        if n > 50 and seed % 5 == 0:
            c[10:40] = c[10]
        if n > 50 and seed % 7 == 0:
            c[5, 0] = np.nan; c[17, 2] = np.inf
        if os.environ.get('VERBOSE'): print(f'seed {seed} n {n} leaf {leaf} kind {kind}', flush=True)
        finite = np.isfinite(c[:, :3]).all(axis=1)   # (PCL skips non-finite points of a non-dense cloud; the oracle takes clean input)
        got, ref = ctx.voxel_grid(c, leaf), oracle.voxel_grid(c[finite], leaf)
        ok = got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32), )
        if not ok and got.shape == c.shape:   # the pass-through (index overflow) case returns the input as it is
            ok = np.array_equal(got, c, equal_nan=True) and np.array_equal(oracle.voxel_grid(c[finite], leaf), c[finite])
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed}: n {n} leaf {leaf} kind {kind} shapes {got.shape} {ref.shape}", flush=True)
print(f"voxel campaign: {N - bad}/{N} clouds bit-identical to the oracle in {time.time() - t0:.0f} s", flush=True)
