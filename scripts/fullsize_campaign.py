"""Dev tool: BASELINE-size pairs (200k x 200k, 10 iterations + fitness) against the CPU oracle, bit level."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import oracle
from icpslam_amd import Context, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
exact = 0
t0 = time.time()
with Context(0) as ctx:
    for seed in range(first, last):
        src, tgt, _ = synth.make_pair(200000, 200000, seed=seed)
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.set_source(src); ctx.set_target(tgt)
        got = ctx.align(want_fitness=True)
        ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=10), want_fitness=True)
        same = (np.array_equal(got["T"].view(np.uint32), np.asarray(ref["T"], np.float32).view(np.uint32)) and got["iterations"] == ref["iterations"]
                and got["n_corr"] == ref["n_corr"])
        exact += int(same)
        print(f"seed {seed}: iterations {got['iterations']}/{ref['iterations']} n_corr {got['n_corr']}/{ref['n_corr']} "
              f"|dT| {np.abs(got['T'].astype(np.float64) - ref['T']).max():.2e} fitness rel diff {abs(got['fitness']-ref['fitness'])/ref['fitness']:.1e} {'bit-identical' if same else 'DIFFERS'}", flush=True)
print(f"200k x 200k pairs {first}..{last}: {exact} of {last-first} bit-identical, {time.time()-t0:.0f} s")
