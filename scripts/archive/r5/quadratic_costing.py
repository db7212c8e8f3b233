"""Round 5, OFFLINE (CPU, the C oracle): how far does GICP move when the inner objective is the exact QUADRATIC form (transformed
points as real numbers, oracle mode GICP_SUMS_SMOOTH) instead of PCL's (points transformed in float32)?  Yardsticks on the same
pairs: PCL's sequential sums against the exact sums, and PCL's loop run backwards against forwards (a pure re-ordering).
Pairs: the random pairs of tests/test_gpu_gicp.py::test_gicp_vs_pcl_ordered_evaluation, and voxel-filtered 200k-point scans of
a drive (the reference's pipeline)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import oracle
from icpslam_amd import synth

R_TOL, T_TOL = 1e-4, 1e-3

def cmp(a, b):
    A, B = np.asarray(a["T"], np.float64), np.asarray(b["T"], np.float64)
    return float(np.abs(A[:3, :3] - B[:3, :3]).max()), float(np.linalg.norm(A[:3, 3] - B[:3, 3]))

def run(src, tgt, gate, mode, iters=10):
    return oracle.icp_align(src, tgt, oracle.default_params(method=oracle.GICP, max_iterations=iters, max_correspondence_distance=gate, gicp_sums=mode))

def random_pair(seed):
    rng = np.random.default_rng(90_000 + seed)
    n_s, n_t = int(rng.integers(3_000, 12_000)), int(rng.integers(3_000, 12_000))
    gate = float(rng.choice([0.5, 1.0, 2.0]))
    src, tgt, _ = synth.make_pair(n_s, n_t, seed=seed)
    return src, tgt, gate

def pipeline_pair(k, scans):
    return scans[k + 1], scans[k], 1.0

def study(name, pairs):
    def one(args):
        src, tgt, gate = args
        return [run(src, tgt, gate, m) for m in (oracle.GICP_SUMS_EXACT, oracle.GICP_SUMS_SMOOTH, oracle.GICP_SUMS_SEQUENTIAL, oracle.GICP_SUMS_SEQUENTIAL_REVERSED)]
    t0 = time.time()
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, pairs))
    print(f"## {name}: {len(pairs)} pairs ({time.time() - t0:.0f} s)")
    for label, i, j in (("quadratic (smooth) vs exact sums", 1, 0), ("quadratic (smooth) vs PCL-ordered", 1, 2), ("exact sums vs PCL-ordered", 0, 2),
                        ("PCL-ordered vs the same loop backwards", 3, 2)):
        d = [cmp(r[i], r[j]) for r in res]
        ok = sum(dR <= R_TOL and dt <= T_TOL for dR, dt in d)
        same = sum(r[i]["iterations"] == r[j]["iterations"] for r in res)
        dts = np.array([x[1] for x in d]); dRs = np.array([x[0] for x in d])
        print(f"{label:42s}: {ok}/{len(d)} within 1e-4 / 1e-3 m | median dR {np.median(dRs):.1e} dt {np.median(dts):.1e} m | 90th {np.quantile(dRs, .9):.1e} {np.quantile(dts, .9):.1e} | worst {dRs.max():.1e} {dts.max():.1e} | same outer iterations {same}")
    sys.stdout.flush()

if __name__ == "__main__":
    n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_scan = int(sys.argv[2]) if len(sys.argv) > 2 else 31
    study("random pairs (seeds 1000..)", [random_pair(s) for s in range(1000, 1000 + n_rand)])
    if n_scan > 1:
        scene = synth.make_scene(7)
        scans = []
        for k in range(n_scan):
            pose = synth.pose_matrix(0.35 * k, 0.02 * k, 0.0, 0.0, 0.0, 0.004 * k)
            raw = synth.scan(scene, pose, 200_000, seed=100 + k)
            scans.append(oracle.voxel_grid(raw, 0.2))
        print("# drive: voxel-filtered scans of", [len(s) for s in scans[:4]], "... points")
        study("reference pipeline pairs (VoxelGrid 0.2 m of 200k-point scans, consecutive scans of a drive)", [pipeline_pair(k, scans) for k in range(n_scan - 1)])
