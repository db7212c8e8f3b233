"""Dev tool: whole alignments (nn_quad_kernel + previous-neighbour bound + fitness completion) against the CPU oracle over
many synthetic scan pairs.  Usage: python scripts/align_campaign.py FIRST LAST"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import oracle
from icpslam_amd import Context, synth

first, last = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
bad = 0
exact = 0
worst = [0.0, 0.0]
t0 = time.time()
with Context(0) as ctx:
    for seed in range(first, last):
        rng = np.random.default_rng(50_000 + seed)
        n_s, n_t = int(rng.integers(33_000, 60_000)), int(rng.integers(20_000, 80_000))
        gate = float(rng.choice([0.3, 1.0, 2.0]))
        iters = int(rng.choice([5, 10, 30]))
        src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
        ctx.set_params(ctx.default_params(), max_iterations=iters, max_correspondence_distance=gate)
        ctx.set_source(src); ctx.set_target(tgt)
        got = ctx.align(want_fitness=True)
        ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=iters, max_correspondence_distance=gate), want_fitness=True)
        dR = float(np.abs(got["T"][:3, :3] - ref["T"][:3, :3]).max()); dt = float(np.linalg.norm(got["T"][:3, 3] - ref["T"][:3, 3]))
        ok = (got["iterations"] == ref["iterations"] and got["n_corr"] == ref["n_corr"] and dR <= 1e-4 and dt <= 1e-3
              and got["converged"] == ref["converged"] and abs(got["fitness"] - ref["fitness"]) <= 1e-6 * max(1.0, ref["fitness"]))
        exact += int(dR == 0.0 and dt == 0.0 and got["iterations"] == ref["iterations"] and got["n_corr"] == ref["n_corr"])
        worst = [max(worst[0], dR), max(worst[1], dt)]
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed}: n {n_s}x{n_t} gate {gate} iters {got['iterations']}/{ref['iterations']} n_corr {got['n_corr']}/{ref['n_corr']} dR {dR:.2e} dt {dt:.2e} "
                  f"fitness {got['fitness']:.6g}/{ref['fitness']:.6g}", flush=True)
print(f"alignments {first}..{last}: {bad} beyond tolerance, {exact} of {last-first} bit-identical to the oracle, worst dR {worst[0]:.2e} dt {worst[1]:.2e}, {time.time()-t0:.0f} s")
