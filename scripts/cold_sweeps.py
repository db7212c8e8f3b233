"""Dev tool: only the FIRST sweep of an alignment (max_iterations = 1), or the k-th and later ones, repeated -- for PMC runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icpslam_amd import Context, synth, NN_GRID
ns, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "200000x200000").split("x"))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
src, tgt, _ = synth.make_pair(ns, nt, seed=4)
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), max_iterations=iters, force_iterations=1, nn_mode=NN_GRID)
    ctx.set_source(src); ctx.set_target(tgt)
    for _ in range(reps):
        r = ctx.align()
    print(r["iterations"], r["n_corr"])
