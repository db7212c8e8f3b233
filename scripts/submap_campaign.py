"""Dev tool: config-3-shaped pairs (200k scan vs 1M-point submap, <= 30 iterations) against the CPU oracle, bit level."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import oracle
from icpslam_amd import Context, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
exact = 0
with Context(0) as ctx:
    for seed in range(first, last):
        src, tgt, _ = synth.make_scan_vs_submap(200000, 1000000, seed=seed)
        ctx.set_params(ctx.default_params(), max_iterations=30)
        ctx.set_source(src); ctx.set_target(tgt)
        got = ctx.align(want_fitness=True)
        ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=30), want_fitness=True)
        same = (np.array_equal(got["T"].view(np.uint32), np.asarray(ref["T"], np.float32).view(np.uint32)) and got["iterations"] == ref["iterations"]
                and got["n_corr"] == ref["n_corr"])
        exact += int(same)
        print(f"seed {seed}: iterations {got['iterations']}/{ref['iterations']} n_corr {got['n_corr']}/{ref['n_corr']} |dT| {np.abs(got['T'].astype(np.float64) - ref['T']).max():.2e} "
              f"{'bit-identical' if same else 'DIFFERS'}", flush=True)
print(f"200k x 1M pairs {first}..{last}: {exact} of {last-first} bit-identical")
