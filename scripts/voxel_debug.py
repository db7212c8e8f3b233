"""Dev tool: ICPGPU_VOXEL_DEBUG=1 phase stamps of the voxel filter's group kernel on a raw scan / uniform cloud."""
import os, sys
os.environ["ICPGPU_VOXEL_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
with Context(0) as ctx:
    scene = synth.make_scene(7)
    c = synth.scan(scene, np.eye(4), 200000, seed=11)
    rng = np.random.default_rng(1)
    u = np.ones((200000, 4), np.float32); u[:, :3] = rng.uniform(-40, 40, (200000, 3)).astype(np.float32)
    for name, cloud, leaf in (("scan", c, 0.2), ("scan", c, 0.2), ("scan", c, 0.5), ("uniform", u, 0.2), ("uniform", u, 0.2)):
        print(name, leaf, flush=True)
        sys.stderr.flush()
        ctx.voxel_grid(cloud, leaf)
