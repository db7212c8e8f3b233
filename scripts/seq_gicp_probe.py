"""Dev tool: config-5-shaped odometry with the reference's literal solver (GICP) next to point-to-point ICP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, P2P_SVD, sequence, synth
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 61
rng = np.random.default_rng(5)
scene = synth.make_scene(5, extent=120.0)
poses = [np.eye(4)]
for _ in range(n_scans - 1):
    poses.append(poses[-1] @ synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3))))
scans = [synth.scan(scene, P, 50000, seed=5000 + k) for k, P in enumerate(poses)]
with Context(0) as ctx:
    for name, method, iters in (("P2P, 10 it (odometer constants)", P2P_SVD, 10), ("P2P, 100 it", P2P_SVD, 100), ("GICP, 10 outer it", GICP, 10)):
        ctx.set_params(ctx.default_params(), method=method, max_iterations=iters)
        t0 = time.perf_counter()
        graph, recs = sequence.run_odometry(ctx, scans)
        dt = time.perf_counter() - t0
        end = np.array(graph.pose(graph.num_poses - 1)[0])
        acc = sum(r["accepted"] for r in recs)
        print(f"{name:32s}: {dt*1e3:8.1f} ms = {(n_scans-1)/dt:7.0f} pairs/s, accepted {acc}/{n_scans-1}, end point {np.linalg.norm(end - poses[-1][:3,3]):6.2f} m from truth "
              f"after {np.linalg.norm(poses[-1][:3,3]):.1f} m, mean its {np.mean([r['iterations'] for r in recs]):.1f}", flush=True)
