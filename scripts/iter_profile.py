"""Dev tool: grid-search kernel time of each of the first ITERS ICP iterations from the identity start (differences of
k-iteration runs, every launch timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth, NN_GRID

ns, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "200000x200000").split("x"))
iters = int(os.environ.get("ITERS", "10"))
with Context(0) as ctx:
    src, tgt, _ = synth.make_scan_vs_submap(ns, nt, seed=3) if nt > 300000 else synth.make_pair(ns, nt, seed=4)
    ctx.profile_sampling(1)
    ctx.set_source(src); ctx.set_target(tgt)
    prev, out = 0.0, []
    for k in range(1, iters + 1):
        ctx.set_params(ctx.default_params(), max_iterations=k, force_iterations=1, nn_mode=NN_GRID)
        ctx.align(); ctx.profile_reset()
        for _ in range(5): ctx.align()
        p = ctx.profile()
        tot = p.grid_ms / 5 * 1e3
        out.append(tot - prev); prev = tot
    print(f"{ns}x{nt} per-iteration NN kernel us:", " ".join(f"{v:.0f}" for v in out), f"| sum {prev:.0f}", flush=True)
