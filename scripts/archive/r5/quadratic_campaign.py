"""Round 5 (GPU): the QUADRATIC inner solver over many registrations.
(1) random pairs (the family of tests/test_gpu_gicp.py): GPU quadratic against the oracle's restatement of the same objective
    (GICP_SUMS_SMOOTH), against PCL's evaluation (GICP_SUMS_SEQUENTIAL), and the GPU's EXACT mode against the same PCL evaluation as
    the yardstick; (2) the reference's pipeline on a drive (VoxelGrid 0.2 m + GICP on 200k-point scans): GPU quadratic against GPU
    exact, scan by scan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import oracle
from icpslam_amd import Context, GICP, GICP_INNER_EXACT, GICP_INNER_QUADRATIC, synth

R_TOL, T_TOL = 1e-4, 1e-3
n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_scan = int(sys.argv[2]) if len(sys.argv) > 2 else 61


def cmp(a, b):
    A, B = np.asarray(a["T"], np.float64), np.asarray(b["T"], np.float64)
    return float(np.abs(A[:3, :3] - B[:3, :3]).max()), float(np.linalg.norm(A[:3, 3] - B[:3, 3]))


def pair(seed):
    rng = np.random.default_rng(90_000 + seed)
    n_s, n_t = int(rng.integers(3_000, 12_000)), int(rng.integers(3_000, 12_000))
    gate = float(rng.choice([0.5, 1.0, 2.0]))
    src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
    return src, tgt, gate


def line(label, d, same=None):
    dR, dt = np.array([x[0] for x in d]), np.array([x[1] for x in d])
    ok = int(np.sum((dR <= R_TOL) & (dt <= T_TOL)))
    tight = int(np.sum((dR <= 1e-7) & (dt <= 1e-6)))
    extra = f" | same outer iterations {same}" if same is not None else ""
    print(f"{label:46s}: {ok}/{len(d)} within 1e-4 / 1e-3 m, {tight} within 1e-7 / 1e-6 m | median dt {np.median(dt):.1e} m, 90th {np.quantile(dt, .9):.1e}, worst dR {dR.max():.1e} dt {dt.max():.1e}{extra}", flush=True)


seeds = list(range(3000, 3000 + n_rand))
def ref(seed):
    src, tgt, gate = pair(seed)
    return [oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_iterations=10, max_correspondence_distance=gate, gicp_sums=m))
            for m in (oracle.GICP_SUMS_SMOOTH, oracle.GICP_SUMS_SEQUENTIAL)]
t0 = time.time()
with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
    refs = list(ex.map(ref, seeds))
print(f"# {n_rand} random pairs (seeds 3000..), oracle runs {time.time() - t0:.0f} s")
q_smooth, q_pcl, e_pcl, q_e, it_same = [], [], [], [], 0
with Context(0) as ctx:
    for seed, (smooth, seq) in zip(seeds, refs):
        src, tgt, gate = pair(seed)
        out = {}
        for inner in (GICP_INNER_QUADRATIC, GICP_INNER_EXACT):
            ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10, max_correspondence_distance=gate, gicp_inner=inner)
            ctx.set_source(src); ctx.set_target(tgt)
            out[inner] = ctx.align()
        q, e = out[GICP_INNER_QUADRATIC], out[GICP_INNER_EXACT]
        q_smooth.append(cmp(q, smooth)); q_pcl.append(cmp(q, seq)); e_pcl.append(cmp(e, seq)); q_e.append(cmp(q, e))
        it_same += q["iterations"] == smooth["iterations"]
line("GPU quadratic vs oracle SMOOTH (same objective)", q_smooth, it_same)
line("GPU quadratic vs PCL-ordered evaluation", q_pcl)
line("GPU exact vs PCL-ordered evaluation (yardstick)", e_pcl)
line("GPU quadratic vs GPU exact", q_e)

if n_scan > 3:
    rng = np.random.default_rng(5)
    scene = synth.make_scene(5, extent=120.0)
    poses = [np.eye(4)]
    for _ in range(n_scan - 1):
        poses.append(poses[-1] @ synth.pose_matrix(0.25, 0.0, 0.0, 0.0, 0.0, np.deg2rad(rng.uniform(-3, 3))))
    scans = [synth.scan(scene, P, 200000, seed=7000 + k) for k, P in enumerate(poses)]
    res = {}
    for inner in (GICP_INNER_EXACT, GICP_INNER_QUADRATIC):
        with Context(0) as ctx:
            ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10, gicp_inner=inner)
            ctx.set_source_voxel_filtered(scans[0], 0.2); ctx.promote_source_to_target()
            out = []
            t0 = time.perf_counter()
            for k in range(1, n_scan):
                ctx.set_source_voxel_filtered(scans[k], 0.2)
                out.append(ctx.align(want_fitness=True))
                ctx.promote_source_to_target()
            res[inner] = (out, (n_scan - 1) / (time.perf_counter() - t0))
    ex, qu = res[GICP_INNER_EXACT], res[GICP_INNER_QUADRATIC]
    print(f"# drive of {n_scan} raw 200k-point scans, VoxelGrid 0.2 m + GICP per scan: exact {ex[1]:.0f} scans/s, quadratic {qu[1]:.0f} scans/s (first scans included)")
    line("pipeline: GPU quadratic vs GPU exact", [cmp(a, b) for a, b in zip(qu[0], ex[0])], sum(a["iterations"] == b["iterations"] for a, b in zip(qu[0], ex[0])))
    fit = np.array([abs(a["fitness"] - b["fitness"]) / b["fitness"] for a, b in zip(qu[0], ex[0])])
    print(f"  fitness scores: relative difference median {np.median(fit):.1e}, worst {fit.max():.1e}; accepted by the odometer's gate (converged and fitness < 20): "
          f"{sum(a['converged'] and a['fitness'] < 20 for a in qu[0])} vs {sum(b['converged'] and b['fitness'] < 20 for b in ex[0])} of {n_scan - 1}")
    # the chained pose (what the odometer publishes)
    def chain(rs):
        P = np.eye(4)
        for r in rs:
            P = P @ np.asarray(r["T"], np.float64)
        return P
    # against the ground truth of the synthetic drive (scan k's pose in scan k-1's frame)
    truth = [np.linalg.inv(poses[k - 1]) @ poses[k] for k in range(1, n_scan)]
    for name, rs in (("exact", ex[0]), ("quadratic", qu[0])):
        err = np.array([np.linalg.norm(np.asarray(r["T"], np.float64)[:3, 3] - t[:3, 3]) for r, t in zip(rs, truth)])
        Pt = np.eye(4)
        for t in truth:
            Pt = Pt @ t
        print(f"  {name:9s} vs ground truth: per-scan translation error median {np.median(err) * 1e3:.2f} mm, 90th {np.quantile(err, .9) * 1e3:.2f} mm; "
              f"end point after the drive {np.linalg.norm(chain(rs)[:3, 3] - Pt[:3, 3]) * 1e3:.1f} mm from the truth")
    Pe, Pq = chain(ex[0]), chain(qu[0])
    print(f"  chained pose after {n_scan - 1} scans ({np.linalg.norm(Pe[:3, 3]):.1f} m travelled): |end point difference| {np.linalg.norm(Pe[:3, 3] - Pq[:3, 3]) * 1e3:.2f} mm, rotation entries {np.abs(Pe[:3, :3] - Pq[:3, :3]).max():.1e}")
