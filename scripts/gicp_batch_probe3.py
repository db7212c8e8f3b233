"""Dev tool: GICP through icpgpu_align_batch by thread count: pairs/s and what an evaluation costs each worker."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth
n, n_pairs = 20000, 32
pairs = [synth.make_pair(n, n, seed=300 + k)[:2] for k in range(4)]
srcs = [pairs[k % 4][0] for k in range(n_pairs)]; tgts = [pairs[k % 4][1] for k in range(n_pairs)]
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    ctx.align_batch(srcs[:8], tgts[:8])
    best = 1e9
    for rep in range(3):
        ctx.profile_reset()
        t0 = time.perf_counter(); res = ctx.align_batch(srcs, tgts); dt = time.perf_counter() - t0
        if dt < best:
            best = dt; p = ctx.profile()
    print(f"threads={os.environ.get('ICPGPU_BATCH_THREADS','auto')}: {n_pairs} GICP pairs of {n}: {best*1e3:.1f} ms = {n_pairs/best:.0f} pairs/s; "
          f"evaluations {p.gicp_cost_launches} at {p.gicp_eval_ms/max(p.gicp_cost_launches,1)*1e3:.2f} us each (summed over workers {p.gicp_eval_ms:.1f} ms), "
          f"covariance passes {p.gicp_cov_launches} at {p.gicp_cov_ms/max(p.gicp_cov_launches,1)*1e3:.0f} us, search {p.grid_ms:.1f} ms in {p.grid_launches}, grid builds {p.grid_builds} {p.grid_build_ms:.1f} ms host, adopted {p.grid_adopted}", flush=True)
