import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from icpslam_amd import Context, synth
from scipy.spatial import cKDTree
for n, seed in ((50000, 2), (200000, 4)):
    src, tgt, Tgt = synth.make_pair(n, n, seed=seed)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params()); ctx.set_source(src); ctx.set_target(tgt)
        r = ctx.align(); ctx.profile_reset(); ctx.fitness(); p = ctx.profile()
        print(n, "fallback points in one fitness sweep:", p.grid_fallback_points, "brute launches", p.nn_launches)
        q = (src[:, :3].astype(np.float64) @ r["T"][:3, :3].T.astype(np.float64) + r["T"][:3, 3])
        d, _ = cKDTree(tgt[:, :3]).query(q)
        print("   NN distance > 1 m:", (d > 1).sum(), " > 4 m:", (d > 4).sum(), " > 10 m:", (d > 10).sum(), " max", d.max())
