"""Dev tool: alignments with the grid search as shipped or on the matrix cores (ICPGPU_TILE_SEARCH=1, icp_tile.hip): results to
an .npz for comparison (python scripts/tile_check.py out.npz), and time per alignment."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
out = {}
gc.disable()
with Context(0) as ctx:
    for name, ns, nt, seed, iters, forced in (("50k", 50000, 50000, 11, 30, 1), ("200k", 200000, 200000, 4, 10, 1), ("200k free", 200000, 200000, 5, 30, 0),
                                              ("scan-submap", 200000, 1000000, 3, 30, 0), ("30k", 30000, 26000, 7, 20, 1)):
        src, tgt, _ = synth.make_scan_vs_submap(ns, nt, seed=seed) if nt > 300000 else synth.make_pair(ns, nt, seed=seed)
        if name == "30k":
            src = src.copy(); src[5, :3] = np.nan; src[6, :3] = (500.0, 500.0, 50.0)
        ctx.set_params(ctx.default_params(), max_iterations=iters, force_iterations=forced)
        ctx.set_source(src); ctx.set_target(tgt)
        r = ctx.align(want_fitness=True)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); r2 = ctx.align(); ts.append(time.perf_counter() - t0)
        out[name + "/T"] = r["T"]; out[name + "/n"] = np.array([r["n_corr"], r["iterations"], int(r["converged"])]); out[name + "/fit"] = np.array([r["fitness"]])
        print(f"{name}: {r['iterations']} it, n_corr {r['n_corr']}, fitness {r['fitness']:.6e}; median {np.median(ts)*1e3:.3f} ms per alignment = {r2['iterations']/np.median(ts):.0f} it/s", flush=True)
np.savez(sys.argv[1], **out)
