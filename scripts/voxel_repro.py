"""Dev tool: one campaign seed of scripts/voxel_campaign.py, repeated, with the differing voxels listed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from icpslam_amd import Context

seed = int(sys.argv[1])
rng = np.random.default_rng(9000 + seed)
n = int(rng.integers(1, 120000))
leaf = float(rng.choice([0.03, 0.1, 0.2, 0.35, 0.77, 2.0, 5.0]))
assert seed % 4 == 0
c = np.ones((n, 4), np.float32)
scale = float(rng.choice([2.0, 30.0, 300.0]))
c[:, :3] = rng.normal(0, scale, (n, 3)).astype(np.float32)
ref = oracle.voxel_grid(c, leaf)
print(f"seed {seed}: n {n} leaf {leaf} scale {scale} -> {len(ref)} voxels")
inv = np.float32(1.0 / leaf)
ijk = np.floor(c[:, :3] * inv).astype(np.int64)
mn = ijk.min(0); d = ijk.max(0) - mn + 1
key = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * d[0] + (ijk[:, 2] - mn[2]) * d[0] * d[1]
ukeys, counts = np.unique(key, return_counts=True)
print("dims", d, "ncells", int(d.prod()), "cpb", (int(d.prod()) + 8191) // 8192, "max voxel", counts.max())
with Context(0) as ctx:
    for rep in range(6):
        got = ctx.voxel_grid(c, leaf)
        same = got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        print("rep", rep, "identical" if same else "DIFFERENT", flush=True)
        if not same and got.shape == ref.shape:
            rows = np.nonzero((got.view(np.uint32) != ref.view(np.uint32)).any(axis=1))[0]
            for r in rows[:6]:
                members = np.nonzero(key == ukeys[r])[0]
                print("   voxel", r, "key", ukeys[r], "members", members[:12], "count", counts[r], "got", got[r], "ref", ref[r])
