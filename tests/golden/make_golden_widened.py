"""Generates the fixtures of the widened rows: tests/golden/rows_f/{voxel_2k,map_3scans,gicp_1k5}.npz.
Run here: python tests/golden/make_golden_widened.py

Like make_golden.py these are data only (inputs + expected outputs); the reference has no vectors for these steps either.
  * voxel_2k   -- pcl::VoxelGrid (icp_odometer.cpp:96-101): expected output from the NumPy restatement in THIS file
                  (float32 arithmetic spelled out), i.e. independent of oracle/icp_oracle.c.
  * map_3scans -- the mapper's map (octree_mapper.cpp:55-90): expected map after each insertion and the nn cloud from the
                  NumPy restatement in THIS file (dictionary of voxels, brute-force float64-free nearest neighbour with
                  the float32 contract of DESIGN.md section 3), independent of oracle/map_oracle.c.
  * gicp_1k5   -- GICP (icp_odometer.cpp:188): expected transform / iterations / covariances from the NumPy restatement
                  oracle/gicp_oracle_np.py (SciPy kd-tree, LAPACK SVD and inverse, PCL's sequential float64 sums), which
                  was written independently of oracle/gicp_oracle.c; this script refuses to write the fixture unless the C
                  restatement (in its PCL-ordered mode) lands within the BASELINE tolerance of it.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from icpslam_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rows_f")
f32 = np.float32


def transform_f32(cloud, T):
    """p = T * s with the a6 contract: fma(m2, z, fma(m1, y, fma(m0, x, m3))) in float32 (emulated through float64:
    a float32 product is exact in float64 and one rounding per fma is what the cast back does)."""
    T = np.asarray(T, f32)
    out = np.ones((cloud.shape[0], 4), f32)
    x, y, z = (cloud[:, k].astype(np.float64) for k in range(3))
    for r in range(3):
        m = T[r].astype(np.float64)
        a = (m[0] * x + m[3]).astype(f32).astype(np.float64)
        a = (m[1] * y + a).astype(f32).astype(np.float64)
        out[:, r] = (m[2] * z + a).astype(f32)
    return out


def voxel_grid_np(cloud, leaf):
    """pcl::VoxelGrid<PointXYZ>::filter with leaf (L, L, L): float32 throughout, centroids summed in point order."""
    pts = cloud[:, :3].astype(f32)
    inv = f32(1.0) / f32(leaf)
    mn, mx = pts.min(0), pts.max(0)
    minb = np.floor(mn * inv).astype(np.int64)
    maxb = np.floor(mx * inv).astype(np.int64)
    div = maxb - minb + 1
    ijk = np.floor(pts * inv).astype(np.int64) - minb
    cell = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    out = []
    for c in np.unique(cell):                       # ascending cell index
        acc = np.zeros(3, f32)
        members = np.flatnonzero(cell == c)         # ascending point index
        for i in members:
            acc = (acc + pts[i]).astype(f32)
        out.append(np.append((acc / f32(len(members))).astype(f32), f32(1.0)))
    return np.array(out, f32)


def first_box_min(p, res):
    """PCL's first octree box (adoptBoundingBoxToPoint on an empty octree, then getKeyBitSize): p +- res / 2 becomes a one-level
    tree, 2 voxels wide, centred on it.  Returns the minimum corner = the lattice origin (double)."""
    lo, hi = p.astype(np.float64) - res / 2, p.astype(np.float64) + res / 2
    side = 2.0 * res                                  # max(ceil(extent / res), 2) voxels = 2 -> depth 1
    return lo - (side - (hi - lo)) / 2.0


class MapNp:
    """First point per voxel of the lattice anchored at the octree's first box minimum (first point - res: PCL's getKeyBitSize); insertion order."""

    def __init__(self, res):
        self.res, self.origin, self.vox, self.pts = float(res), None, set(), []

    def add(self, cloud, pose):
        moved = transform_f32(cloud, pose)
        added = 0
        for p in moved:
            if not np.isfinite(p[:3]).all():
                continue
            if self.origin is None:
                self.origin = first_box_min(p[:3], self.res)
            k = tuple(np.floor((p[:3].astype(np.float64) - self.origin) / self.res).astype(np.int64))
            if k not in self.vox:
                self.vox.add(k)
                self.pts.append(p.copy())
                added += 1
        return added

    def points(self):
        return np.array(self.pts, f32).reshape(-1, 4)

    def nn_cloud(self, cloud, pose, pose_inv):
        m = self.points()
        q = transform_f32(cloud, pose)
        sel = []
        for p in q:
            if not np.isfinite(p[:3]).all():
                continue
            d = (m[:, :3] - p[:3]).astype(f32)                       # float32 differences
            dx, dy, dz = (d[:, k].astype(np.float64) for k in range(3))
            d2 = (dx * dx).astype(f32).astype(np.float64)            # dx*dx, then two fmas
            d2 = (dy * dy + d2).astype(f32).astype(np.float64)
            d2 = (dz * dz + d2).astype(f32)
            sel.append(int(np.argmin(d2)))                           # first minimum = lowest index
        return transform_f32(m[sel], pose_inv)


def main():
    # f2
    src, _, _ = synth.make_pair(2000, 10, seed=91)
    np.savez_compressed(os.path.join(OUT, "voxel_2k.npz"), cloud=src, leaf=f32(0.5), expected=voxel_grid_np(src, 0.5),
                        expected_small_leaf=voxel_grid_np(src, 0.05))
    # f4
    scene = synth.make_scene(92)
    poses = [synth.pose_matrix(0.7 * k, 0.1 * k, 0.0, 0.0, 0.0, 0.03 * k).astype(f32) for k in range(3)]
    scans = [synth.scan(scene, P, 1500, seed=920 + k) for k, P in enumerate(poses)]
    scans[1][3, :3] = np.nan
    m = MapNp(0.5)
    added, sizes = [], []
    for s, P in zip(scans, poses):
        added.append(m.add(s, P))
        sizes.append(len(m.pts))
    probe = synth.scan(scene, poses[2], 600, seed=929)
    probe[5, :3] = np.inf
    pinv = np.linalg.inv(poses[2].astype(np.float64)).astype(f32)
    np.savez_compressed(os.path.join(OUT, "map_3scans.npz"), scans=np.array(scans), poses=np.array(poses), resolution=0.5,
                        added=np.array(added), sizes=np.array(sizes), map_points=m.points(), probe=probe, probe_pose=poses[2],
                        probe_pose_inv=pinv, nn_cloud=m.nn_cloud(probe, poses[2], pinv))
    print("map", added, sizes, "nn", m.nn_cloud(probe, poses[2], pinv).shape)
    # f1
    from oracle import gicp_oracle_np as gnp
    a, b, _ = synth.make_pair(1500, 1500, seed=93)
    r = gnp.gicp_align(a, b, sums="sequential")
    cov = gnp.covariances(b)
    c = oracle.icp_align(a, b, oracle.default_params(method=oracle.GICP, gicp_sums=oracle.GICP_SUMS_SEQUENTIAL), want_fitness=True)
    dR = np.abs(r["T"][:3, :3].astype(np.float64) - c["T"][:3, :3]).max()
    dt = np.linalg.norm(r["T"][:3, 3].astype(np.float64) - c["T"][:3, 3])
    assert dR <= 1e-4 and dt <= 1e-3 and r["n_corr"] == c["n_corr"], (dR, dt)
    assert np.abs(cov - oracle.gicp_covariances(b, pcl_order=True)).max() <= 1e-9
    # fitness (mean squared NN distance after the transform) with the float32 contract, brute force in NumPy
    moved = gnp.transform_f32(a, r["T"])
    d = (b[None, :, :3] - moved[:, None, :]).astype(np.float64)
    d2 = (d[..., 0] * d[..., 0]).astype(f32).astype(np.float64)
    d2 = (d[..., 1] * d[..., 1] + d2).astype(f32).astype(np.float64)
    d2 = (d[..., 2] * d[..., 2] + d2).astype(f32)
    fitness = float(d2.min(axis=1).astype(np.float64).sum() / a.shape[0])
    np.savez_compressed(os.path.join(OUT, "gicp_1k5.npz"), src=a, tgt=b, T=r["T"], converged=r["converged"],
                        iterations=r["iterations"], n_corr=r["n_corr"], fitness=fitness, cov_tgt=cov)
    print("gicp (NumPy restatement)", r["iterations"], r["converged"], r["n_corr"], "C restatement:", c["iterations"],
          "dR %.1e dt %.1e" % (dR, dt), "fitness", fitness, c["fitness"])


if __name__ == "__main__":
    main()
