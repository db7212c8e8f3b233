"""Dev tool: whole GICP registrations (resident evaluation server and all) against the CPU oracle over many synthetic pairs.
Usage: python scripts/gicp_campaign.py FIRST LAST"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import oracle
from icpslam_amd import Context, GICP, synth

first, last = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
bad = 0
worst = [0.0, 0.0]
t0 = time.time()
with Context(0) as ctx:
    for seed in range(first, last):
        rng = np.random.default_rng(90_000 + seed)
        n_s, n_t = int(rng.integers(3_000, 12_000)), int(rng.integers(3_000, 12_000))
        gate = float(rng.choice([0.5, 1.0, 2.0]))
        src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10, max_correspondence_distance=gate)
        ctx.set_source(src); ctx.set_target(tgt)
        got = ctx.align(want_fitness=True)
        ref = oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_iterations=10, max_correspondence_distance=gate),
                               want_fitness=True)
        dR = float(np.abs(got["T"][:3, :3] - ref["T"][:3, :3]).max()); dt = float(np.linalg.norm(got["T"][:3, 3] - ref["T"][:3, 3]))
        worst = [max(worst[0], dR), max(worst[1], dt)]
        # (the tolerances of tests/test_gpu_gicp.py: the delta < 1 stop sits on a 1e-6 m threshold)
        ok = (abs(got["iterations"] - ref["iterations"]) <= 1 and dR <= 1e-4 and dt <= 1e-3 and got["converged"] == ref["converged"]
              and abs(got["fitness"] - ref["fitness"]) <= 1e-3 * max(1.0, ref["fitness"]))
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed}: n {n_s}x{n_t} gate {gate} iters {got['iterations']}/{ref['iterations']} conv {got['converged']}/{ref['converged']} "
                  f"dR {dR:.2e} dt {dt:.2e} fitness {got['fitness']:.6g}/{ref['fitness']:.6g}", flush=True)
print(f"GICP registrations {first}..{last}: {bad} mismatches, worst dR {worst[0]:.2e} dt {worst[1]:.2e}, {time.time()-t0:.0f} s")
