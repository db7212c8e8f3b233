"""Dev tool: a handful of aligns at one size/mode, for rocprofv3 runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth, NN_GRID, NN_BRUTE
ns, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "200000x200000").split("x"))
mode = NN_BRUTE if (len(sys.argv) > 2 and sys.argv[2] == "brute") else NN_GRID
src, tgt, _ = synth.make_pair(ns, nt, seed=4)
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1, nn_mode=mode)
    ctx.set_source(src); ctx.set_target(tgt)
    for _ in range(3):
        r = ctx.align(want_fitness=True)
    print(r["iterations"], r["n_corr"])
