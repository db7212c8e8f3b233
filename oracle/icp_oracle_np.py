"""Second, independent restatement of PCL point-to-point ICP in NumPy/SciPy (float64 solve, cKDTree NN).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: exists to cross-check
oracle/icp_oracle.c.  Restates SURVEY.md Appendix A.1 for the call sites at
/root/reference/src/icpslam/icp_odometer.cpp:188-201 and src/icpslam/octree_mapper.cpp:104-117.
Deliberately written differently from the C oracle (library SVD, two-pass demeaned covariance,
library kd-tree with float64 distances) so a shared bug is unlikely.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

NOT_CONVERGED, ITERATIONS, TRANSFORM, ABS_MSE, REL_MSE, NO_CORRESPONDENCES = range(6)


def transform_cloud_f32(T32: np.ndarray, xyz: np.ndarray) -> np.ndarray:
    """p = R s + t evaluated in float32 (rounding differs from the fmaf chain by <= 2 ulp)."""
    R, t = T32[:3, :3].astype(np.float32), T32[:3, 3].astype(np.float32)
    return (xyz.astype(np.float32) @ R.T + t).astype(np.float32)


def umeyama(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Eigen::umeyama(src=p, dst=q, with_scaling=false), float64."""
    mp, mq = p.mean(axis=0), q.mean(axis=0)
    sigma = (q - mq).T @ (p - mp) / p.shape[0]
    U, d, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1
    R = U @ np.diag(S) @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mq - R @ mp
    return T


def icp_align(src, tgt, max_iterations=10, transformation_epsilon=1e-6, max_correspondence_distance=1.0,
              euclidean_fitness_epsilon=-np.finfo(np.float64).max, min_correspondences=3, guess=None,
              force_iterations=False, want_fitness=False):
    src = np.asarray(src, np.float32)[:, :3]
    tgt = np.asarray(tgt, np.float32)[:, :3]
    out = dict(T=np.eye(4, dtype=np.float32), converged=False, iterations=0, state=NOT_CONVERGED, n_corr=0,
               mse=0.0, fitness=float("nan"), trace=[])
    if tgt.shape[0] == 0:
        return out
    tree = cKDTree(tgt.astype(np.float64), leafsize=15)
    final = np.eye(4) if guess is None else np.asarray(guess, np.float64).copy()
    r2 = max_correspondence_distance ** 2
    mse_prev = np.finfo(np.float64).max
    nr, converged, state, n_c, mse = 0, False, NOT_CONVERGED, 0, 0.0
    while True:
        X = transform_cloud_f32(final.astype(np.float32), src)
        if X.shape[0]:
            d, j = tree.query(X.astype(np.float64), k=1)
            d2 = (d * d)
        else:
            d2, j = np.zeros(0), np.zeros(0, np.int64)
        keep = d2.astype(np.float32) <= r2
        n_c = int(keep.sum())
        if n_c < min_correspondences:
            state, converged = NO_CORRESPONDENCES, False
            break
        p = X[keep].astype(np.float64)
        q = tgt[j[keep]].astype(np.float64)
        Tk = umeyama(p, q)
        final = Tk @ final
        mse = float(d2[keep].mean())
        out["trace"].append(dict(Tk=Tk.copy(), final=final.copy(), n_corr=n_c, mse=mse))
        nr += 1
        if nr >= max_iterations:
            converged, state = True, ITERATIONS
        elif not force_iterations:
            cos_angle = 0.5 * (np.trace(Tk[:3, :3]) - 1.0)
            tsq = float(Tk[:3, 3] @ Tk[:3, 3])
            if cos_angle >= 1.0 - transformation_epsilon and tsq <= transformation_epsilon:
                converged, state = True, TRANSFORM
            elif abs(mse - mse_prev) < 1e-12:
                converged, state = True, ABS_MSE
            elif abs(mse - mse_prev) / mse_prev < euclidean_fitness_epsilon:
                converged, state = True, REL_MSE
            mse_prev = mse
        if converged:
            break
    out.update(T=final.astype(np.float32), converged=converged, iterations=nr, state=state, n_corr=n_c, mse=mse)
    if want_fitness and src.shape[0]:
        X = transform_cloud_f32(final.astype(np.float32), src)
        d, _ = tree.query(X.astype(np.float64), k=1)
        out["fitness"] = float((d * d).mean())
    return out
