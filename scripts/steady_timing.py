"""Dev tool: grid-mode ICP iteration cost (a) from the identity start (b) at the converged transform."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth, NN_GRID

sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(200000, 200000)]
with Context(0) as ctx:
    for ns, nt in sizes:
        src, tgt, _ = synth.make_scan_vs_submap(ns, nt, seed=3) if nt > 300000 else synth.make_pair(ns, nt, seed=4)
        order = os.environ.get("SRC_ORDER", "")
        if order:  # experiment: spatially coherent source order (cells of `order` metres, z-major like a voxel filter's output)
            off = src[:, :3].min(0) - float(order) if os.environ.get("SRC_PHASE") else 0.0
            c = np.floor((src[:, :3] - off) / float(order)).astype(np.int64)
            if os.environ.get("SRC_INNER") == "x":  # within a cell: ascending x instead of the (shuffled) input order
                src = src[np.argsort(src[:, 0], kind="stable")]
                c = np.floor((src[:, :3] - off) / float(order)).astype(np.int64)
            c -= c.min(0)
            d = c.max(0) + 1
            src = src[np.argsort((c[:, 2] * d[1] + c[:, 1]) * d[0] + c[:, 0], kind="stable")]
        ctx.set_params(ctx.default_params(), max_iterations=40, nn_mode=NN_GRID)
        ctx.set_source(src); ctx.set_target(tgt)
        Tc = ctx.align()["T"]
        ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1, nn_mode=NN_GRID)
        for name, guess in (("from identity", None), ("at convergence", Tc)):
            ctx.align(guess=guess); ctx.profile_reset()
            t0 = time.perf_counter()
            for _ in range(5): ctx.align(guess=guess)
            wall = (time.perf_counter() - t0) / 5
            p = ctx.profile()
            print(f"{ns}x{nt} {name:15s}: NN kernel {p.grid_ms/max(1,p.grid_timed)*1e3:7.1f} us/iter, align(10) {wall*1e3:7.3f} ms", flush=True)
