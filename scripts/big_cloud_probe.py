import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from icpslam_amd import Context, synth, NN_GRID, NN_BRUTE
rng = np.random.default_rng(0)
n = 3000000
base = synth.make_pair(200000, 200000, seed=4)[1]
# 3M points: the 200k scan replicated with jitter (a very dense cloud)
tgt = np.repeat(base, 15, axis=0).copy(); tgt[:, :3] += rng.normal(0, 0.03, (n, 3)).astype(np.float32)
T = synth.pose_matrix(0.2, -0.1, 0.02, 0.0, 0.01, 0.02)
src = tgt[rng.permutation(n)[:2000000]].copy()
src[:, :3] = (src[:, :3].astype(np.float64) @ np.linalg.inv(T.astype(np.float64))[:3, :3].T + np.linalg.inv(T.astype(np.float64))[:3, 3]).astype(np.float32)
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), max_iterations=20, nn_mode=NN_GRID)
    ctx.set_source(src); ctx.set_target(tgt)
    t0 = time.perf_counter(); r = ctx.align(want_fitness=True); dt = time.perf_counter() - t0
    print("2M x 3M align:", round(dt * 1e3, 1), "ms, iterations", r["iterations"], "n_corr", r["n_corr"], "fitness", r["fitness"])
    print("recovered:", np.abs(r["T"] - T).max())
    idx, d2 = ctx.nn(r["T"])
    print("nn ok", (idx >= 0).mean())
