"""Dev tool: per-iteration cost of the ICP loop in brute / grid mode on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth, NN_BRUTE, NN_GRID

sizes = [(5000, 5000), (50000, 50000), (200000, 200000)]
if len(sys.argv) > 1:
    sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
modes = [(NN_GRID, "grid")] + ([(NN_BRUTE, "brute")] if os.environ.get("WITH_BRUTE") else [])
with Context(0) as ctx:
    for (ns, nt) in sizes:
        if nt > 300000:
            src, tgt, _ = synth.make_scan_vs_submap(ns, nt, seed=3)
        else:
            src, tgt, _ = synth.make_pair(ns, nt, seed=4)
        for mode, name in modes:
            ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1, nn_mode=mode)
            ctx.set_source(src); ctx.set_target(tgt)
            ctx.align()
            ctx.profile_reset()
            n = 5
            t0 = time.perf_counter()
            for _ in range(n):
                r = ctx.align()
            wall = (time.perf_counter() - t0) / n
            p = ctx.profile()
            k_ms = (p.grid_ms / max(1, p.grid_timed)) if mode == NN_GRID else (p.nn_ms / max(1, p.nn_timed))
            print(f"{name:5s} {ns}x{nt}: align(10 it) {wall*1e3:8.3f} ms -> {10/wall:9.1f} it/s | NN kernel {k_ms*1e3:8.1f} us "
                  f"reduce {p.reduce_ms/max(1,p.reduce_launches)*1e3:6.1f} us n_corr {r['n_corr']}", flush=True)
        ctx.set_params(ctx.default_params(), nn_mode=NN_GRID)
        ctx.set_source(src + np.float32(0)); ctx.set_target(tgt + np.float32(0))   # new clouds -> rebuild
        ctx.profile_reset(); t0 = time.perf_counter(); ctx.align(want_fitness=True); w = time.perf_counter() - t0; p = ctx.profile()
        print(f"      index build {p.grid_build_ms*1e3:.1f} us; fitness fallback points {p.grid_fallback_points}; full odometer call {w*1e3:.2f} ms", flush=True)
