"""Dev tool: GICP mode timing (GPU vs the CPU oracle) at a few sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from icpslam_amd import Context, synth, GICP
sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(5000, 5000), (50000, 50000), (200000, 200000)]
with Context(0) as ctx:
    for ns, nt in sizes:
        src, tgt, _ = synth.make_pair(ns, nt, seed=4)
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
        ctx.set_source(src); ctx.set_target(tgt)
        ctx.profile_reset()
        t0 = time.perf_counter(); r = ctx.align(want_fitness=True); cold = time.perf_counter() - t0
        p = ctx.profile()
        t0 = time.perf_counter(); r = ctx.align(want_fitness=True); warm = time.perf_counter() - t0
        line = (f"GICP {ns}x{nt}: first call {cold*1e3:8.2f} ms (covariances {p.gicp_cov_ms:.2f} ms for 2 clouds), repeat {warm*1e3:8.2f} ms; "
                f"outer its {r['iterations']}, cost evals {p.gicp_cost_launches}, n_corr {r['n_corr']}")
        if ns <= 50000:
            t0 = time.perf_counter(); ref = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP), want_fitness=True); cpu = time.perf_counter() - t0
            line += f" | CPU oracle {cpu*1e3:.0f} ms, dT {np.abs(r['T']-ref['T']).max():.2e}"
        print(line, flush=True)
