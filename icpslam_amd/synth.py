"""Deterministic synthetic LIDAR scan generator.

Surface-structured "street scene" ray-cast on a 64-ring Velodyne pattern.  Used by the
parity tests, the golden-fixture script and bench.py; contains no ICP arithmetic.

All clouds are float32 N x 4 (x, y, z, 1.0f): the in-memory layout of ``pcl::PointXYZ``
that the reference hands to PCL at /root/reference/src/icpslam/icp_odometer.cpp:193-194.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

GROUND_Z = -1.73
MAX_RANGE = 80.0
MIN_RANGE = 2.0
N_RINGS = 64
ELEV_MIN_DEG = -24.8
ELEV_MAX_DEG = 2.0
RANGE_SIGMA = 0.02


@dataclass
class Scene:
    facade_y: tuple          # (+y plane, -y plane)
    boxes: np.ndarray        # (nb, 6) xmin,ymin,zmin,xmax,ymax,zmax
    cyls: np.ndarray         # (nc, 4) cx, cy, r, ztop


def make_scene(seed: int, extent: float = 60.0) -> Scene:
    rng = np.random.default_rng(seed)
    fy = (8.0 + rng.uniform(0, 4), -(8.0 + rng.uniform(0, 4)))
    nb = int(rng.integers(20, 41))
    boxes = np.zeros((nb, 6))
    for i in range(nb):
        cx = rng.uniform(-extent, extent)
        cy = rng.uniform(fy[1] + 1.5, fy[0] - 1.5)
        if abs(cx) < 3.0 and abs(cy) < 3.0:      # keep the sensor's immediate surroundings free
            cx += 6.0 * (1 if cx >= 0 else -1)
        lx, ly = (4.0, 1.8) if rng.uniform() < 0.7 else (1.8, 4.0)
        boxes[i] = (cx - lx / 2, cy - ly / 2, GROUND_Z, cx + lx / 2, cy + ly / 2, GROUND_Z + 1.5)
    nc = 20
    cyls = np.zeros((nc, 4))
    for i in range(nc):
        cx = rng.uniform(-extent, extent)
        cy = rng.uniform(fy[1] + 0.5, fy[0] - 0.5)
        if abs(cx) < 3.0 and abs(cy) < 3.0:
            cx += 6.0 * (1 if cx >= 0 else -1)
        cyls[i] = (cx, cy, rng.uniform(0.15, 0.4), GROUND_Z + rng.uniform(3.0, 8.0))
    return Scene(fy, boxes, cyls)


def _ground_height(x, y):
    return GROUND_Z + 0.1 * np.sin(0.3 * x) * np.cos(0.25 * y)


def _raycast(scene: Scene, origin: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """Return hit points (M,3) in world frame and a validity mask."""
    M = dirs.shape[0]
    ox, oy, oz = origin
    dx, dy, dz = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    t_best = np.full(M, np.inf)
    kind = np.zeros(M, dtype=np.int8)   # 1 = ground (gets the undulation applied afterwards)

    with np.errstate(divide="ignore", invalid="ignore"):
        # ground plane
        t = (GROUND_Z - oz) / dz
        ok = (dz < 0) & (t > 0)
        upd = ok & (t < t_best)
        t_best = np.where(upd, t, t_best)
        kind = np.where(upd, 1, kind)

        # facades with window recesses (0.5 m deep, 1 m wide every 3 m, z in [-0.5, 1.5])
        for Y in scene.facade_y:
            t = (Y - oy) / dy
            ok = (t > 0) & np.isfinite(t)
            hx = ox + t * dx
            hz = oz + t * dz
            recess = ok & (np.mod(hx, 3.0) < 1.0) & (hz > -0.5) & (hz < 1.5)
            Y2 = Y + (0.5 if Y > 0 else -0.5)
            t2 = (Y2 - oy) / dy
            t = np.where(recess, t2, t)
            hz = oz + t * dz
            ok = ok & (hz < 12.0) & (hz > GROUND_Z - 0.5)
            upd = ok & (t < t_best)
            t_best = np.where(upd, t, t_best)
            kind = np.where(upd, 2, kind)

        # boxes (slab method)
        inv = 1.0 / dirs
        for b in scene.boxes:
            t1 = (b[0:3][None, :] - origin[None, :]) * inv
            t2 = (b[3:6][None, :] - origin[None, :]) * inv
            tmin = np.nanmax(np.minimum(t1, t2), axis=1)
            tmax = np.nanmin(np.maximum(t1, t2), axis=1)
            ok = (tmax >= tmin) & (tmin > 0)
            upd = ok & (tmin < t_best)
            t_best = np.where(upd, tmin, t_best)
            kind = np.where(upd, 3, kind)

        # vertical cylinders
        a = dx * dx + dy * dy
        for c in scene.cyls:
            fx, fy_ = ox - c[0], oy - c[1]
            bb = 2 * (fx * dx + fy_ * dy)
            cc = fx * fx + fy_ * fy_ - c[2] * c[2]
            disc = bb * bb - 4 * a * cc
            sq = np.sqrt(np.where(disc >= 0, disc, 0.0))
            t = (-bb - sq) / (2 * a)
            hz = oz + t * dz
            ok = (disc >= 0) & (t > 0) & (hz < c[3]) & (hz > GROUND_Z - 0.2)
            upd = ok & (t < t_best)
            t_best = np.where(upd, t, t_best)
            kind = np.where(upd, 4, kind)

    valid = np.isfinite(t_best) & (t_best >= MIN_RANGE) & (t_best <= MAX_RANGE)
    t_safe = np.where(valid, t_best, 0.0)
    return t_safe, valid, kind


def pose_matrix(tx, ty, tz, roll, pitch, yaw) -> np.ndarray:
    """4x4 float64 pose, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = (tx, ty, tz)
    return T


def scan(scene: Scene, pose: np.ndarray, n_points: int, seed: int, noise: float = RANGE_SIGMA,
         oversample: float = 1.6) -> np.ndarray:
    """One scan of ``n_points`` returns taken at ``pose`` (sensor->world), in the SENSOR frame."""
    rng = np.random.default_rng(seed)
    if n_points == 0:
        return np.zeros((0, 4), np.float32)
    while True:
        steps = max(8, int(math.ceil(n_points * oversample / N_RINGS)))
        elev = np.deg2rad(np.linspace(ELEV_MIN_DEG, ELEV_MAX_DEG, N_RINGS))
        az = (np.arange(steps) + rng.uniform()) * (2 * math.pi / steps)
        ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
        d_s = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], se * np.ones_like(az)[None, :]],
                       axis=-1).reshape(-1, 3)
        R, t = pose[:3, :3], pose[:3, 3]
        d_w = d_s @ R.T
        rng_t, valid, kind = _raycast(scene, t, d_w)
        if int(valid.sum()) >= n_points:
            break
        oversample *= 1.5
    hit = t[None, :] + rng_t[:, None] * d_w
    g = kind == 1
    hit[g, 2] = _ground_height(hit[g, 0], hit[g, 1])
    # express in the sensor frame, then add range noise along the (sensor-frame) ray
    p_s = (hit - t[None, :]) @ R
    sel = np.flatnonzero(valid)
    sel = sel[rng.permutation(sel.size)[:n_points]]
    p = p_s[sel]
    if noise > 0:
        r = np.linalg.norm(p, axis=1, keepdims=True)
        p = p * (1.0 + rng.normal(0.0, noise, size=(p.shape[0], 1)) / np.maximum(r, 1e-9))
    out = np.ones((n_points, 4), np.float32)
    out[:, :3] = p.astype(np.float32)
    return out


def random_motion(rng: np.random.Generator) -> np.ndarray:
    """Small motion of Appendix D.3: |t_xy| <= 0.5 m, |t_z| <= 0.05 m, yaw <= 3 deg, roll/pitch <= 0.5 deg."""
    return pose_matrix(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.05, 0.05),
                       math.radians(rng.uniform(-0.5, 0.5)), math.radians(rng.uniform(-0.5, 0.5)),
                       math.radians(rng.uniform(-3.0, 3.0)))


def make_pair(n_src: int, n_tgt: int, seed: int, noise: float = RANGE_SIGMA):
    """(source, target, T_gt): source = scan from pose B in B's frame, target = scan from pose A = I.

    T_gt (float64 4x4) maps source -> target like the transform the reference chains at
    /root/reference/src/icpslam/icp_odometer.cpp:112-113.
    """
    rng = np.random.default_rng(seed)
    scene = make_scene(int(rng.integers(1 << 31)))
    T_b = random_motion(rng)
    s_seed, t_seed = int(rng.integers(1 << 31)), int(rng.integers(1 << 31))
    target = scan(scene, np.eye(4), n_tgt, t_seed, noise)
    source = scan(scene, T_b, n_src, s_seed, noise)
    return source, target, T_b


def make_known_answer_pair(n: int, seed: int):
    """Noise-free family: target = T_gt * source (same points, permuted). ICP must recover T_gt."""
    rng = np.random.default_rng(seed)
    scene = make_scene(int(rng.integers(1 << 31)))
    src = scan(scene, np.eye(4), n, int(rng.integers(1 << 31)), noise=0.0)
    T = random_motion(rng)
    # keep the motion well inside the basin: scale translation/rotation down
    T[:3, 3] *= 0.2
    tgt = np.ones_like(src)
    tgt[:, :3] = (src[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    tgt = tgt[rng.permutation(n)]
    return src, tgt, T


def make_submap(n_points: int, seed: int, n_poses: int = 5, spacing: float = 1.0, dedup: float = 0.05):
    """C3 target: union of scans from ``n_poses`` poses ``spacing`` m apart, de-duplicated, exactly n_points.

    Returns (submap in world frame, scene, list of poses).
    """
    rng = np.random.default_rng(seed)
    scene = make_scene(int(rng.integers(1 << 31)))
    per = int(math.ceil(n_points * 1.35 / n_poses))
    clouds, poses = [], []
    for k in range(n_poses):
        P = pose_matrix(spacing * k, 0.0, 0.0, 0.0, 0.0, math.radians(rng.uniform(-2, 2)))
        c = scan(scene, P, per, int(rng.integers(1 << 31)))
        w = c[:, :3].astype(np.float64) @ P[:3, :3].T + P[:3, 3]
        clouds.append(w)
        poses.append(P)
    allp = np.concatenate(clouds, axis=0)
    key = np.floor(allp / dedup).astype(np.int64)
    key = (key[:, 0] * 73856093) ^ (key[:, 1] * 19349663) ^ (key[:, 2] * 83492791)
    _, first = np.unique(key, return_index=True)
    allp = allp[np.sort(first)]
    allp = allp[rng.permutation(allp.shape[0])]
    if allp.shape[0] < n_points:   # pad by re-sampling with sub-voxel jitter
        extra = allp[rng.integers(0, allp.shape[0], n_points - allp.shape[0])] + rng.normal(0, 0.01, (n_points - allp.shape[0], 3))
        allp = np.concatenate([allp, extra], axis=0)
    out = np.ones((n_points, 4), np.float32)
    out[:, :3] = allp[:n_points].astype(np.float32)
    return out, scene, poses


def make_scan_vs_submap(n_scan: int, n_map: int, seed: int):
    """C3 pair: source = new scan near the middle pose (sensor frame, perturbed), target = world submap."""
    submap, scene, poses = make_submap(n_map, seed)
    rng = np.random.default_rng(seed + 7919)
    P = poses[len(poses) // 2] @ random_motion(rng)
    src = scan(scene, P, n_scan, int(rng.integers(1 << 31)))
    # the mapper hands ICP a cloud already moved by the raw pose estimate; emulate a small residual
    resid = random_motion(rng)
    resid[:3, 3] *= 0.5
    guess_pose = P @ np.linalg.inv(resid)
    moved = np.ones_like(src)
    moved[:, :3] = (src[:, :3].astype(np.float64) @ guess_pose[:3, :3].T + guess_pose[:3, 3]).astype(np.float32)
    # ground truth that maps `moved` onto the map: P * guess_pose^-1
    T_gt = P @ np.linalg.inv(guess_pose)
    return moved, submap, T_gt
