"""Dev tool: cost of getFitnessScore after an alignment (one ungated sweep + completion + reduction)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth

with Context(0) as ctx:
    for n in (22000, 50000, 200000):
        src, tgt, _ = synth.make_pair(n, n, seed=4)
        ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1)
        ctx.set_source(src); ctx.set_target(tgt)
        for _ in range(3): ctx.align(want_fitness=True)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps): ctx.align()
        t1 = time.perf_counter()
        for _ in range(reps): ctx.align(want_fitness=True)
        t2 = time.perf_counter()
        ctx.profile_reset()
        r = ctx.align(want_fitness=True)
        p = ctx.profile()
        print(f"{n}: align {(t1-t0)/reps*1e3:.3f} ms, align+fitness {(t2-t1)/reps*1e3:.3f} ms -> fitness {(t2-2*t1+t0)/reps*1e3:.3f} ms; "
              f"points completed by brute force {p.grid_fallback_points}, fitness {r['fitness']:.5f}", flush=True)
