"""Dev tool: GICP through icpgpu_align_batch at 20k and 50k points per cloud, by thread count and server variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth
for n, n_pairs in ((20000, 32), (50000, 32)):
    pairs = [synth.make_pair(n, n, seed=300 + k)[:2] for k in range(4)]
    srcs = [pairs[k % 4][0] for k in range(n_pairs)]; tgts = [pairs[k % 4][1] for k in range(n_pairs)]
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
        ctx.align_batch(srcs[:8], tgts[:8])
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); res = ctx.align_batch(srcs, tgts); best = min(best, time.perf_counter() - t0)
        same = all(np.array_equal(res[k]["T"], res[k % 4]["T"]) for k in range(n_pairs))
        print(f"threads={os.environ.get('ICPGPU_BATCH_THREADS','auto')} resident_max={os.environ.get('ICPGPU_GICP_RESIDENT_MAX','default')}: "
              f"{n_pairs} GICP pairs of {n}: {best*1e3:.1f} ms = {n_pairs/best:.0f} pairs/s, identical results for identical pairs: {same}", flush=True)
