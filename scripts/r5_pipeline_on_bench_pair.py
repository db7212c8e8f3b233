import os, sys, time
sys.path.insert(0, "/root/repo")
if os.environ.get("WITH_TORCH"): import torch
import numpy as np
from icpslam_amd import Context, GICP, synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
raw = [a, b]
with Context(0) as ctx:
    if os.environ.get("BIG_FIRST"):
        ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1)
        ctx.set_source(a); ctx.set_target(b)
        for _ in range(5): ctx.align(want_fitness=True)
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    ctx.set_source_voxel_filtered(raw[1], 0.2); ctx.promote_source_to_target()
    for k in range(4):
        ctx.set_source_voxel_filtered(raw[k % 2], 0.2); ctx.align(want_fitness=True); ctx.promote_source_to_target()
    ctx.profile_reset()
    t0 = time.perf_counter(); N = 50
    for k in range(N):
        ctx.set_source_voxel_filtered(raw[k % 2], 0.2); ctx.align(want_fitness=True); ctx.promote_source_to_target()
    dt = time.perf_counter() - t0
    p = ctx.profile()
    print(f"torch={bool(os.environ.get('WITH_TORCH'))} big_first={bool(os.environ.get('BIG_FIRST'))}: {N/dt:.0f} scans/s, {dt/N*1e3:.2f} ms per scan, evaluations {p.gicp_cost_launches/N:.0f} per scan at {p.gicp_eval_ms/max(1,p.gicp_cost_launches)*1e3:.2f} us, device solves {p.gicp_device_solves}, host solves {p.gicp_host_solves}, choice {p.gicp_solver_choice}")
