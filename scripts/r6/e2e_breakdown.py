"""Dev tool (round 6): where a scan pair of bench.py's e2e loop (set_source from a host buffer + index build + 10 forced point-to-point
iterations + fitness + promote, 200k x 200k) spends its time beyond the resident alignment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from icpslam_amd import Context, synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
clouds = (a, b)
N = 60
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1)
    ctx.set_source(clouds[1]); ctx.promote_source_to_target()
    for k in range(4):
        ctx.set_source(clouds[k % 2]); ctx.align(want_fitness=True); ctx.promote_source_to_target()
    ctx.profile_reset()
    t = dict(put=0.0, align=0.0, promote=0.0)
    for k in range(N):
        t0 = time.perf_counter(); ctx.set_source(clouds[k % 2]); t1 = time.perf_counter()
        ctx.align(want_fitness=True); t2 = time.perf_counter()
        ctx.promote_source_to_target(); t3 = time.perf_counter()
        t["put"] += t1 - t0; t["align"] += t2 - t1; t["promote"] += t3 - t2
    p = ctx.profile()
    print(f"e2e loop, us per pair: set_source (H2D 3.2 MB) {t['put']/N*1e6:.0f} | align (index build + 10 sweeps + fitness) {t['align']/N*1e6:.0f} | promote {t['promote']/N*1e6:.1f}"
          f" | grid builds {p.grid_builds/N:.2f} per pair at {p.grid_build_ms/max(p.grid_builds,1)*1e3:.0f} us host wall each")
    ctx.set_source(a); ctx.set_target(b); ctx.align(want_fitness=True)
    t0 = time.perf_counter()
    for _ in range(N): ctx.align(want_fitness=True)
    print(f"resident pair: align + fitness {(time.perf_counter()-t0)/N*1e6:.0f} us")
