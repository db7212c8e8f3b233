"""Dev tool: f4 timings -- map growth and nn-cloud construction at BASELINE config 3 sizes (200k scan vs ~1M-point map)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
from icpslam_amd.mapper import OctreeMapper
from icpslam_amd.sequence import pose_from_matrix

res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cpu = len(sys.argv) > 3 and sys.argv[3] == "cpu"
scene = synth.make_scene(seed=3)
poses = [synth.pose_matrix(1.0 * k, 0.05 * k, 0.0, 0.0, 0.0, 0.01 * k) for k in range(n_scans)]
scans = [synth.scan(scene, P, 200000, seed=300 + k) for k, P in enumerate(poses)]
with Context(0) as ctx:
    ctx.profile_sampling(1)   # dev tool: time every sweep
    m = OctreeMapper(ctx, octree_resolution=res)
    for k, (s, P) in enumerate(zip(scans, poses)):
        ctx.profile_reset()
        t0 = time.perf_counter()
        ok, tr, refined, info = m.refineTransformAndGrowMap(s, pose_from_matrix(P))
        wall = time.perf_counter() - t0
        p = ctx.profile()
        it = info.get("icp", {}).get("iterations", 0) if not info.get("seeded") else 0
        print(f"scan {k}: map {m.map_size:8d} (+{info['added']:6d})  step {wall*1e3:7.2f} ms | insert {p.map_insert_ms:6.3f} ms, "
              f"nn cloud {p.map_nn_ms:6.3f} ms, icp {it:2d} it, grid kernel {p.grid_ms:6.3f} ms / {p.grid_launches} launches, "
              f"grid builds {p.grid_builds} ({p.grid_build_ms:5.2f} ms)", flush=True)
if cpu:
    import oracle
    ref = oracle.VoxelMap(res)
    for k, (s, P) in enumerate(zip(scans[:3], poses[:3])):
        t0 = time.perf_counter(); ref.add_points(s, P); t1 = time.perf_counter()
        nn = ref.nn_cloud(s, P, np.linalg.inv(P.astype(np.float64)).astype(np.float32)); t2 = time.perf_counter()
        print(f"cpu oracle scan {k}: insert {1e3*(t1-t0):8.1f} ms, nn cloud {1e3*(t2-t1):8.1f} ms, map {len(ref)}")
