"""Dev tool: floor of the grid kernel -- source == target (every query is settled by its own copy in the octant stage)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth, NN_GRID
src, tgt, _ = synth.make_pair(200000, 200000, seed=4)
with Context(0) as ctx:
    for name, s, t in (("pair", src, tgt), ("self", tgt, tgt)):
        for leaf in (0.0, 0.2):
            ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1, nn_mode=NN_GRID)
            if leaf:
                s2, t2 = ctx.voxel_grid(s, leaf), ctx.voxel_grid(t, leaf)
            else:
                s2, t2 = s, t
            ctx.set_source(s2); ctx.set_target(t2)
            ctx.align(); ctx.profile_reset()
            for _ in range(5): ctx.align()
            p = ctx.profile()
            print(f"{name} leaf={leaf}: n_s={len(s2)} n_t={len(t2)} NN kernel {p.grid_ms/max(1,p.grid_timed)*1e3:.1f} us/iter ({p.grid_launches} grid launches, {p.nn_launches} brute)")
