"""Dev tool: icpgpu_align_batch (config 4 shape: 64 pairs of 50k, <= 10 iterations + fitness) against the number of host
threads, several repetitions each (the figure is noisy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
n_pairs, n = 64, 50000
n_distinct = int(os.environ.get("DISTINCT", "8"))   # 64: every pair its own host arrays (102 MB: cold in the CPU caches)
base = [synth.make_pair(n, n, seed=1000 + k)[:2] for k in range(n_distinct)]
srcs = [base[k % n_distinct][0] for k in range(n_pairs)]; tgts = [base[k % n_distinct][1] for k in range(n_pairs)]
for workers in (4, 6, 8):
    os.environ["ICPGPU_BATCH_WORKERS"] = str(workers)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.align_batch(srcs[:8], tgts[:8], want_fitness=True)
        rates = []
        for _ in range(7):
            t0 = time.perf_counter()
            ctx.align_batch(srcs, tgts, want_fitness=True)
            rates.append(n_pairs / (time.perf_counter() - t0))
        rates.sort()
        print(f"workers={workers}: pairs/s min {rates[0]:.0f} median {rates[3]:.0f} max {rates[-1]:.0f}", flush=True)
