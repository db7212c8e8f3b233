"""Dev tool: where a scan of the reference's pipeline (VoxelGrid 0.2 m + GICP, 10 outer iterations) spends its time: wall
clock per stage (host timers around the C-ABI calls) and the device-side counters of the profile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 31
rng = np.random.default_rng(5)
scene = synth.make_scene(5, extent=120.0)
poses = [np.eye(4)]
for _ in range(n_scans - 1):
    poses.append(poses[-1] @ synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3))))
scans = [synth.scan(scene, P, 200000, seed=7000 + k) for k, P in enumerate(poses)]
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    ctx.set_source_voxel_filtered(scans[0], 0.2); ctx.promote_source_to_target()
    for k in (1, 2):
        ctx.set_source_voxel_filtered(scans[k], 0.2); ctx.align(want_fitness=True); ctx.promote_source_to_target()
    ctx.profile_reset()
    t = dict(filter=0.0, align=0.0, promote=0.0)
    its = evals = 0
    t_all = time.perf_counter()
    for k in range(3, n_scans):
        t0 = time.perf_counter(); n = ctx.set_source_voxel_filtered(scans[k], 0.2); t1 = time.perf_counter()
        r = ctx.align(want_fitness=True); t2 = time.perf_counter()
        ctx.promote_source_to_target(); t3 = time.perf_counter()
        t["filter"] += t1 - t0; t["align"] += t2 - t1; t["promote"] += t3 - t2
        its += r["iterations"]
    wall = time.perf_counter() - t_all
    p = ctx.profile(); m = n_scans - 3
    print(f"{m} scans of 200k -> ~{n} points: {wall/m*1e3:.2f} ms per scan = {m/wall:.0f} scans/s; outer iterations {its/m:.1f}, cost evaluations {p.gicp_cost_launches/m:.0f} per scan")
    print(f"  host wall per scan: H2D + bbox + voxel filter {t['filter']/m*1e3:.3f} ms | align (GICP + fitness) {t['align']/m*1e3:.3f} ms | promote {t['promote']/m*1e3:.3f} ms")
    print(f"  device counters per scan: voxel {p.voxel_ms/m*1e3:.0f} us | covariances {p.gicp_cov_ms/m*1e3:.0f} us ({p.gicp_cov_launches/m:.1f} clouds) | correspondence search + Mahalanobis kernels {p.grid_ms*p.grid_launches/max(p.grid_timed,1)/m*1e3:.0f} us ({p.grid_launches/m:.1f} outer iterations' worth + the fitness sweep; {p.grid_timed} of {p.grid_launches} timed)")
