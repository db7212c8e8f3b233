"""Dev tool: for GICP campaign seeds that do not match the oracle bit for bit -- where do the inputs of the optimisation differ?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import oracle
from icpslam_amd import Context, GICP, synth
oracle.build()
with Context(0) as ctx:
    for seed in [int(a) for a in sys.argv[1:]]:
        rng = np.random.default_rng(90_000 + seed)
        n_s, n_t = int(rng.integers(3_000, 12_000)), int(rng.integers(3_000, 12_000))
        gate = float(rng.choice([0.5, 1.0, 2.0]))
        src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10, max_correspondence_distance=gate)
        for name, cloud in (("src", src), ("tgt", tgt)):
            ctx.set_source(cloud)
            got = ctx.gicp_covariances(); ref = oracle.gicp_covariances(cloud)
            d = np.abs(got - ref).reshape(len(cloud), -1).max(1)
            bad = np.flatnonzero(d > 0)
            print(f"seed {seed} {name}: {len(bad)} of {len(cloud)} covariances differ, max {d.max():.3e}", bad[:5], d[bad[:5]])
        for it in (1, 2, 3, 10):
            ctx.set_params(ctx.default_params(), method=GICP, max_iterations=it, max_correspondence_distance=gate)
            ctx.set_source(src); ctx.set_target(tgt)
            g = ctx.align()
            r = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_iterations=it, max_correspondence_distance=gate))
            print(f"   max_iterations {it}: iterations {g['iterations']}/{r['iterations']} n_corr {g['n_corr']}/{r['n_corr']} |dT| {np.abs(g['T'].astype(np.float64)-r['T']).max():.3e}")
