"""Dev tool: the reference odometer's actual per-scan pipeline on raw 200k-point scans -- VoxelGrid(0.2 m)
(icp_odometer.cpp:96-101,177; icpslam.yaml:14) -> GICP, 10 iterations, fitness gate (icp_odometer.cpp:188-201) -> pose
chain + keyframes -- next to the same loop with point-to-point ICP and without the filter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, P2P_SVD, sequence, synth
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 41
rng = np.random.default_rng(5)
scene = synth.make_scene(5, extent=120.0)
poses = [np.eye(4)]
for _ in range(n_scans - 1):
    poses.append(poses[-1] @ synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3))))
scans = [synth.scan(scene, P, 200000, seed=7000 + k) for k, P in enumerate(poses)]
with Context(0) as ctx:
    for name, method, leaf in (("VoxelGrid 0.2 m + GICP (the reference's pipeline)", GICP, 0.2), ("VoxelGrid 0.2 m + P2P", P2P_SVD, 0.2),
                               ("raw 200k + GICP", GICP, None), ("raw 200k + P2P", P2P_SVD, None)):
        ctx.set_params(ctx.default_params(), method=method, max_iterations=10)
        sequence.run_odometry(ctx, scans[:3], voxel_leaf=leaf)      # warm-up
        t0 = time.perf_counter()
        graph, recs = sequence.run_odometry(ctx, scans, voxel_leaf=leaf)
        dt = time.perf_counter() - t0
        end = np.array(graph.pose(graph.num_poses - 1)[0])
        acc = sum(r["accepted"] for r in recs)
        print(f"{name:50s}: {dt*1e3/(n_scans-1):7.2f} ms per scan = {(n_scans-1)/dt:6.0f} scans/s, accepted {acc}/{n_scans-1}, "
              f"{ctx.n_target:6d} points after the filter, end point {np.linalg.norm(end - poses[-1][:3,3]):5.2f} m from truth after "
              f"{np.linalg.norm(poses[-1][:3,3]):.1f} m", flush=True)
