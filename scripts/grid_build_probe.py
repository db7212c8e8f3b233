"""Dev tool: cost of (re)building the target's search grid (set_target + first search), per cloud size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth, NN_GRID

with Context(0) as ctx:
    for n in (22000, 50000, 200000, 1000000):
        src, tgt, _ = synth.make_scan_vs_submap(50000, n, seed=3) if n > 300000 else synth.make_pair(n, n, seed=4)
        tgts = [tgt.copy() for _ in range(6)]
        ctx.set_params(ctx.default_params(), max_iterations=1, force_iterations=1, nn_mode=NN_GRID)
        ctx.set_source(src)
        ctx.set_target(tgts[0]); ctx.align()
        ctx.profile_reset()
        t_set = t_first = 0.0
        for t in tgts[1:]:
            t0 = time.perf_counter(); ctx.set_target(t); t1 = time.perf_counter(); ctx.align(); t2 = time.perf_counter()
            t_set += t1 - t0; t_first += t2 - t1
        ctx.align(); t3 = time.perf_counter(); ctx.align(); t4 = time.perf_counter()
        p = ctx.profile()
        k = len(tgts) - 1
        print(f"target {n:8d}: set_target (H2D) {t_set/k*1e3:.3f} ms, first align(1 it) {t_first/k*1e3:.3f} ms, repeat align {((t4-t3))*1e3:.3f} ms, "
              f"grid builds {p.grid_builds} ({p.grid_build_ms/max(1,p.grid_builds):.3f} ms each, host time)", flush=True)
