"""Dev tool: the result mailbox under repetition -- every value travels as a {value, sequence number} pair in ONE 16-byte
store; a pair seen half-written would show up as a result that differs from the first run's.  Many identical alignments
(point-to-point: ~10 sweeps each; GICP: ~250 evaluations each), every result compared bit for bit with the first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, P2P_SVD, synth

n_p2p = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_gicp = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
with Context(0) as ctx:
    for name, method, size, reps in (("point-to-point 50k x 50k", P2P_SVD, 50000, n_p2p), ("GICP 5k x 5k", GICP, 5000, n_gicp),
                                     ("GICP 40k x 40k (resident server)", GICP, 40000, n_gicp // 4), ("GICP 100k x 100k", GICP, 100000, n_gicp // 10)):
        src, tgt, _ = synth.make_pair(size, size, seed=4)
        ctx.set_params(ctx.default_params(), method=method, max_iterations=10)
        ctx.set_source(src); ctx.set_target(tgt)
        first = ctx.align(want_fitness=True)
        key = (first["T"].tobytes(), first["iterations"], first["n_corr"], first["mse"], first["fitness"])
        bad = 0
        t0 = time.time()
        ctx.profile_reset()
        for _ in range(reps):
            r = ctx.align(want_fitness=True)
            bad += (r["T"].tobytes(), r["iterations"], r["n_corr"], r["mse"], r["fitness"]) != key
        p = ctx.profile()
        print(f"{name}: {reps} alignments ({p.iterations} iterations, {p.gicp_cost_launches} cost evaluations) in {time.time()-t0:.1f} s, "
              f"{bad} results differ from the first", flush=True)
