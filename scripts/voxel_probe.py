"""Dev tool: device time of the voxel filter (f2) on a raw synthetic LIDAR scan and on a uniform cloud (one point per voxel);
run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth

with Context(0) as ctx:
    ctx.profile_sampling(1)
    scene = synth.make_scene(7)
    clouds = {}
    for n in (200000, 50000):
        clouds[f"scan{n}"] = synth.scan(scene, np.eye(4), n, seed=11)
    rng = np.random.default_rng(1)
    u = np.ones((200000, 4), np.float32); u[:, :3] = rng.uniform(-40, 40, (200000, 3)).astype(np.float32)
    clouds["uniform200000"] = u
    for name, c in clouds.items():
        for leaf in (0.2, 0.5):
            out = ctx.voxel_grid(c, leaf); ctx.profile_reset()
            for _ in range(5): out = ctx.voxel_grid(c, leaf)
            p = ctx.profile(); us = p.voxel_ms / p.voxel_launches * 1e3
            print(f"{name:16s} leaf {leaf}: {us:7.1f} us device time, {len(c)} -> {len(out)} points", flush=True)
