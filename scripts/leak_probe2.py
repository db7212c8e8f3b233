"""Dev tool: device + host memory drift over many contexts that use every path with its own scratch (P2P, GICP, voxel filter in
both paths, map, batch)."""
import os, sys, resource
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from icpslam_amd import Context, GICP, P2P_SVD, synth
scene = synth.make_scene(3)
raw = synth.scan(scene, np.eye(4), 60000, seed=1)
src, tgt, _ = synth.make_pair(30000, 30000, seed=1)
dense = np.ones((9000, 4), np.float32); dense[:, :3] = np.random.default_rng(0).uniform(0.01, 0.19, (9000, 3))
torch.cuda.synchronize()
def free(): return torch.cuda.mem_get_info()[0] / 2**20
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
f0, r0 = None, None
for n in range(1, 121):
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), method=GICP if n % 2 else P2P_SVD, max_iterations=5)
        ctx.set_source_voxel_filtered(raw[: 20000 + 300 * n], 0.2); ctx.set_target(ctx.voxel_grid(raw, 0.2)); ctx.align(want_fitness=True)
        ctx.voxel_grid(dense, 0.2)                       # over capacity: the sort path
        ctx.set_params(ctx.default_params(), max_iterations=5)
        ctx.align_batch([src[:9000], src[:8000]], [tgt[:9000], tgt[:7000]], want_fitness=True)
    if n == 20: f0, r0 = free(), rss()
    if n in (40, 80, 120): print(f"{n} contexts: device drift {f0 - free():.1f} MiB, host max-RSS drift {rss() - r0:.1f} MiB since context 20", flush=True)
