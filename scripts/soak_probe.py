"""Dev tool: soak -- many contexts created/destroyed, many aligns with changing cloud sizes; watches device memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from icpslam_amd import Context, GICP, synth
free0 = torch.cuda.mem_get_info()[0]
src, tgt, _ = synth.make_pair(60000, 60000, seed=1)
t0 = time.perf_counter()
for k in range(150):
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params())
        ctx.set_source(src[: 1000 + 300 * k]); ctx.set_target(tgt[: 60000 - 200 * k])
        ctx.align(want_fitness=True)
print(f"150 contexts: {time.perf_counter()-t0:.1f} s, device memory drift {(free0 - torch.cuda.mem_get_info()[0])/2**20:.1f} MiB")
rng = np.random.default_rng(0)
with Context(0) as ctx:
    free1 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    for k in range(3000):
        ns, nt = int(rng.integers(100, 60000)), int(rng.integers(100, 60000))
        ctx.set_params(ctx.default_params(), method=GICP if k % 50 == 0 else 0, max_iterations=int(rng.integers(1, 12)))
        ctx.set_source(src[:ns]); ctx.set_target(tgt[:nt])
        r = ctx.align(want_fitness=(k % 3 == 0))
        if k % 7 == 0: ctx.promote_source_to_target()
        if k % 11 == 0:
            ctx.map_reset(0.5); ctx.map_add_points(src[:ns]); ctx.set_source(tgt[:nt]); ctx.map_nn_target(np.eye(4), np.eye(4), want_cloud=False); ctx.align()
    print(f"3000 mixed aligns: {time.perf_counter()-t0:.1f} s, device memory held by the context {(free1 - torch.cuda.mem_get_info()[0])/2**20:.1f} MiB")
print(f"after everything: drift {(free0 - torch.cuda.mem_get_info()[0])/2**20:.1f} MiB")
