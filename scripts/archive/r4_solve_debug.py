"""Dev probe: one GICP registration through the device solver with diagnostics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ICPGPU_DEBUG"] = "1"
import numpy as np
from icpslam_amd import Context, GICP, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
src, tgt, _ = synth.make_pair(n, n, seed=77)
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    ctx.set_source(src); ctx.set_target(tgt)
    for rep in range(3):
        t0 = time.perf_counter()
        r = ctx.align(want_fitness=True)
        t1 = time.perf_counter()
        p = ctx.profile()
        print(f"rep {rep}: {1e3*(t1-t0):.3f} ms, iterations {r['iterations']}, n_corr {r['n_corr']}, evaluations so far {p.gicp_cost_launches}, eval ms {p.gicp_eval_ms:.3f}", flush=True)
