"""Dev tool (round 5): icpgpu_align_batch at config 4's shape (64 pairs of 50k, <= 10 iterations + fitness) against host threads x
lock-step depth x groups in flight -- how many groups must be in flight for one entry to fill the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
n_pairs, n = int(os.environ.get("PAIRS", "64")), 50000
n_distinct = int(os.environ.get("DISTINCT", "64"))   # 64: config 4's own pairs (seeds 1000+k); fewer: the same pairs again and again
base = [synth.make_pair(n, n, seed=1000 + k)[:2] for k in range(n_distinct)]
srcs = [base[k % n_distinct][0] for k in range(n_pairs)]; tgts = [base[k % n_distinct][1] for k in range(n_pairs)]
combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4, 8, 4), (4, 8, 8), (2, 8, 8), (1, 8, 8), (4, 4, 16), (4, 8, 16), (8, 8, 8), (2, 4, 16), (4, 16, 4), (1, 4, 16)]
for threads, depth, groups in combos:
    os.environ["ICPGPU_BATCH_THREADS"] = str(threads)
    os.environ["ICPGPU_BATCH_DEPTH"] = str(depth)
    os.environ["ICPGPU_BATCH_GROUPS"] = str(groups)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.align_batch(srcs, tgts, want_fitness=True)
        ms = []
        for _ in range(int(os.environ.get("REPS", "12"))):
            t0 = time.perf_counter()
            ctx.align_batch(srcs, tgts, want_fitness=True)
            ms.append(1e3 * (time.perf_counter() - t0))
        first = ms[0]
        ms.sort()
        print(f"threads={threads} depth={depth} groups={groups}: ms per {n_pairs} pairs min {ms[0]:.2f} median {ms[len(ms) // 2]:.2f} max {ms[-1]:.2f} first {first:.2f} -> {n_pairs * 1e3 / ms[len(ms) // 2]:.0f} pairs/s", flush=True)
