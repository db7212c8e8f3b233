import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from icpslam_amd import Context, sequence, synth
rng = np.random.default_rng(5)
scene = synth.make_scene(5, extent=120.0)
poses = [np.eye(4)]
for _ in range(20):
    poses.append(poses[-1] @ synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3))))
scans = [synth.scan(scene, P, 200000, seed=7000 + k) for k, P in enumerate(poses)]
with Context(0) as ctx:
    ctx.set_params(ctx.default_params())
    sequence.run_odometry(ctx, scans, voxel_leaf=0.2)
