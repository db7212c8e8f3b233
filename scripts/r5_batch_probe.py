"""Dev tool (round 5): icpgpu_align_batch at config 4's shape (64 pairs of 50k, <= 10 iterations + fitness) against host threads x
lock-step depth -- how many groups must be in flight for one entry to fill the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
n_pairs, n = 64, 50000
base = [synth.make_pair(n, n, seed=1000 + k)[:2] for k in range(8)]
srcs = [base[k % 8][0] for k in range(n_pairs)]; tgts = [base[k % 8][1] for k in range(n_pairs)]
combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4, 8), (8, 8), (8, 4), (16, 4), (4, 16), (2, 16), (2, 8), (1, 16)]
for threads, depth in combos:
    os.environ["ICPGPU_BATCH_THREADS"] = str(threads)
    os.environ["ICPGPU_BATCH_DEPTH"] = str(depth)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.align_batch(srcs, tgts, want_fitness=True)
        ms = []
        for _ in range(12):
            t0 = time.perf_counter()
            ctx.align_batch(srcs, tgts, want_fitness=True)
            ms.append(1e3 * (time.perf_counter() - t0))
        first = ms[0]
        ms.sort()
        print(f"threads={threads} depth={depth}: ms per 64 pairs min {ms[0]:.2f} median {ms[6]:.2f} max {ms[-1]:.2f} first {first:.2f} -> {64e3 / ms[6]:.0f} pairs/s", flush=True)
