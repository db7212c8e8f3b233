"""Dev tool: GICP through icpgpu_align_batch -- several worker contexts, each with its own resident evaluation server."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth
n_pairs, n = 24, 20000
pairs = [synth.make_pair(n, n, seed=300 + k)[:2] for k in range(6)]
srcs = [pairs[k % 6][0] for k in range(n_pairs)]; tgts = [pairs[k % 6][1] for k in range(n_pairs)]
for workers in (1, 4, 8, 16):
    os.environ["ICPGPU_BATCH_WORKERS"] = str(workers)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
        ctx.align_batch(srcs[:workers], tgts[:workers])
        t0 = time.perf_counter()
        res = ctx.align_batch(srcs, tgts)
        dt = time.perf_counter() - t0
        same = all(np.array_equal(res[k]["T"], res[k % 6]["T"]) for k in range(n_pairs))
        print(f"workers={workers}: {n_pairs} GICP pairs of {n} in {dt*1e3:.1f} ms = {n_pairs/dt:.0f} pairs/s, identical results for identical pairs: {same}", flush=True)
