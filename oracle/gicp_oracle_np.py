"""gicp_oracle_np.py -- SECOND, independently written restatement of pcl::GeneralizedIterativeClosestPoint (PCL 1.8.x),
the solver the reference instantiates at /root/reference/src/icpslam/icp_odometer.cpp:188 and
src/icpslam/octree_mapper.cpp:104.  TEST INFRASTRUCTURE ONLY (see oracle/icp_oracle.h): only tests/ and the fixture
generator tests/golden/make_golden_widened.py may import it.

       ***  PARITY UNPINNED  ***  (PCL is not under /root/reference and not installable here)

Written from the published algorithm (Segal et al., "Generalized-ICP"; PCL's gicp.hpp drives GSL's vector_bfgs2 minimiser
with Fletcher's line search, ported in pcl/registration/bfgs.h), NOT from oracle/gicp_oracle.c: NumPy / SciPy / LAPACK do
the work the C restatement does by hand --
    neighbours        scipy.spatial.cKDTree (float64 distances)          C: own kd-tree, float32 contract distances
    3x3 SVD           numpy.linalg.svd (LAPACK gesdd)                     C: one-sided Jacobi
    Mahalanobis       numpy.linalg.inv (LU)                               C: adjugate
    sums              "sequential": numpy.add.accumulate (PCL's order)    C: plain loop
                      "exact": math.fsum (correctly rounded exact sum)    C: three-fold TwoSum expansion
                      "smooth": exact sums of the objective WITHOUT the float32 rounding of the transformed points (T p in
                      float64 from the float matrix) -- what the GPU's quadratic inner solver minimises; C: ORC_GICP_SUMS_SMOOTH
so the two agree to rounding in every intermediate quantity but not bit for bit, and -- BFGS being a chaotic consumer of
its sums -- their final transforms agree within the BASELINE tolerance (1e-4 in R, 1e-3 m in t), not to the last bit.
That distance is what tests/test_oracle.py::test_gicp_two_restatements_agree and the fixture
tests/golden/rows_f/gicp_1k5.npz measure.

Constants: PCL's constructor defaults the reference leaves untouched (k_correspondences_ 20, gicp_epsilon_ 1e-3,
rotation_epsilon_ 2e-3, max_inner_iterations_ 20, gradient tolerance 1e-2) and the reference's own settings
(icp_odometer.h:63-65: correspondence distance 1.0, transformation epsilon 1e-6, 10 iterations).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np
from scipy.spatial import cKDTree

K_CORR = 20
GICP_EPS = 1e-3
ROT_EPS = 2e-3
MAX_INNER = 20
GRAD_TOL = 1e-2
f32 = np.float32

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.cosf.restype = _libm.sinf.restype = ctypes.c_float
_libm.cosf.argtypes = _libm.sinf.argtypes = [ctypes.c_float]


def _cosf(a):
    return f32(_libm.cosf(float(a)))


def _sinf(a):
    return f32(_libm.sinf(float(a)))


# ---------------------------------------------------------------------------------------------------------------------
# float32 point transform with the a6 contract: fma(m2, z, fma(m1, y, fma(m0, x, m3)))
# ---------------------------------------------------------------------------------------------------------------------
def transform_f32(pts, T):
    """pts (n, >=3) float32, T 4x4 float32 -> (n, 3) float32.  A float32 product is exact in float64 and the cast back is
    the fma's single rounding (double rounding can only bite on an exact float32 midpoint)."""
    T = np.asarray(T, f32).astype(np.float64)
    x, y, z = (pts[:, k].astype(np.float64) for k in range(3))
    out = np.empty((pts.shape[0], 3), f32)
    for r in range(3):
        a = (T[r, 0] * x + T[r, 3]).astype(f32).astype(np.float64)
        a = (T[r, 1] * y + a).astype(f32).astype(np.float64)
        out[:, r] = (T[r, 2] * z + a).astype(f32)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# computeCovariances
# ---------------------------------------------------------------------------------------------------------------------
def covariances(cloud):
    """(n, 3, 3) float64: per point, covariance of its 20 nearest neighbours (itself included), singular values replaced
    by (1, 1, gicp_epsilon)."""
    p32 = np.ascontiguousarray(cloud[:, :3], f32)
    n = p32.shape[0]
    if n < K_CORR:
        raise ValueError("cloud smaller than k_correspondences_")
    _, nbr = cKDTree(p32.astype(np.float64)).query(p32.astype(np.float64), k=K_CORR)
    q = p32[nbr]                                           # (n, 20, 3) float32
    mean = q.astype(np.float64).sum(axis=1) / K_CORR       # sums of 20 float32 values are exact in float64
    # cov(k, l) += pt[k] * pt[l]: a float * float product (rounded to float32) added to a double
    prod = (q[:, :, :, None] * q[:, :, None, :]).astype(np.float64)   # float32 products, then widened
    cov = prod.sum(axis=1) / K_CORR - mean[:, :, None] * mean[:, None, :]
    il = np.tril_indices(3, -1)
    cov[:, il[1], il[0]] = cov[:, il[0], il[1]]            # PCL computes the lower triangle and mirrors it
    U, _, _ = np.linalg.svd(cov)
    v = np.array([1.0, 1.0, GICP_EPS])
    return np.einsum("k,nrk,nck->nrc", v, U, U)


# ---------------------------------------------------------------------------------------------------------------------
# the state vector x = (tx, ty, tz, roll, pitch, yaw)
# ---------------------------------------------------------------------------------------------------------------------
def _quat_mul(a, b):   # Eigen's generic quaternion product, (w, x, y, z), float32 scalars
    return (a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
            a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
            a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1])


def apply_state(T, x):
    """applyState: T.R <- Rz(x5) Ry(x4) Rx(x3) T.R (Eigen float AngleAxis -> quaternion -> matrix), T.t += x[0:3]."""
    T = np.array(T, f32)
    z0 = f32(0)
    half = f32(0.5)
    hx, hy, hz = half * f32(x[3]), half * f32(x[4]), half * f32(x[5])
    qx = (_cosf(hx), _sinf(hx), z0, z0)
    qy = (_cosf(hy), z0, _sinf(hy), z0)
    qz = (_cosf(hz), z0, z0, _sinf(hz))
    w, a, b, c = _quat_mul(_quat_mul(qz, qy), qx)
    two = f32(2)
    tx, ty, tz = two * a, two * b, two * c
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * a, ty * a, tz * a
    tyy, tyz, tzz = ty * b, tz * b, tz * c
    one = f32(1)
    R = np.array([[one - (tyy + tzz), txy - twz, txz + twy],
                  [txy + twz, one - (txx + tzz), tyz - twx],
                  [txz - twy, tyz + twx, one - (txx + tyy)]], f32)
    old = T[:3, :3].copy()
    new = np.zeros((3, 3), f32)
    for r in range(3):
        for c_ in range(3):
            s = f32(0)
            for k in range(3):
                s = f32(s + R[r, k] * old[k, c_])
            new[r, c_] = s
    T[:3, :3] = new
    T[:3, 3] = (T[:3, 3] + np.array(x[:3], f32)).astype(f32)
    return T


def _state_from_matrix(T):
    T = np.asarray(T, np.float64)
    return np.array([T[0, 3], T[1, 3], T[2, 3], math.atan2(T[2, 1], T[2, 2]), math.asin(-T[2, 0]),
                     math.atan2(T[1, 0], T[0, 0])], np.float64)


def _rotation_gradient(x, Rm):
    """computeRDerivative: d/d(roll, pitch, yaw) of Rz Ry Rx contracted with Rm (sum_ij dR(j, i) * Rm(i, j))."""
    phi, theta, psi = x[3], x[4], x[5]
    cphi, sphi, cth, sth, cpsi, spsi = math.cos(phi), math.sin(phi), math.cos(theta), math.sin(theta), math.cos(psi), math.sin(psi)
    dphi = np.array([[0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth],
                     [0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth],
                     [0, cphi * cth, -cth * sphi]])
    dth = np.array([[-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth],
                    [-spsi * sth, cth * sphi * spsi, cphi * cth * spsi],
                    [-cth, -sphi * sth, -cphi * sth]])
    dpsi = np.array([[-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth],
                     [cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth],
                     [0, 0, 0]])
    out = []
    for d in (dphi, dth, dpsi):
        s = 0.0
        for i in range(3):
            for j in range(3):
                s += d[j, i] * Rm[i, j]
        out.append(s)
    return out


class _Cost:
    """OptimizationFunctorWithIndices: f(x) = (1/m) sum r^T M r, r = T(x) p_src - p_tgt; gradient as PCL computes it."""

    def __init__(self, src, tgt, maha, base, sums):
        self.src, self.tgt, self.maha, self.base, self.m = src, tgt, maha, np.asarray(base, f32), src.shape[0]
        self.pb = transform_f32(src, self.base).astype(np.float64)   # base_transformation_ * p_src (rotation gradient)
        # "smooth": the objective WITHOUT the float32 rounding of the transformed points (the C oracle's ORC_GICP_SUMS_SMOOTH, the
        # GPU's quadratic inner solver): T p and base p in float64 from the float matrices, exact sums
        self.smooth = sums == "smooth"
        if self.smooth:
            self.src64 = np.asarray(src, np.float64)[:, :3]
            b64 = self.base.astype(np.float64)
            self.pb = self.src64 @ b64[:3, :3].T + b64[:3, 3]
        self.sum = (lambda a: math.fsum(a.tolist())) if sums in ("exact", "smooth") else (lambda a: float(np.add.accumulate(a)[-1]))
        self.evaluations = 0

    def fdf(self, x):
        self.evaluations += 1
        T = apply_state(self.base, x)
        if self.smooth:
            T64 = np.asarray(T, f32).astype(np.float64)
            res = self.src64 @ T64[:3, :3].T + T64[:3, 3] - np.asarray(self.tgt, np.float64)[:, :3]
        else:
            res = (transform_f32(self.src, T) - self.tgt).astype(np.float64)   # float32 difference, then widened
        M = self.maha
        temp = np.empty_like(res)
        for r in range(3):   # temp = M * res, left to right like the C++ expression
            temp[:, r] = M[:, r, 0] * res[:, 0] + M[:, r, 1] * res[:, 1] + M[:, r, 2] * res[:, 2]
        f = self.sum(res[:, 0] * temp[:, 0] + res[:, 1] * temp[:, 1] + res[:, 2] * temp[:, 2]) / self.m
        g = np.zeros(6)
        sc = 2.0 / self.m
        for r in range(3):
            g[r] = self.sum(temp[:, r]) * sc
        Rm = np.array([[self.sum(self.pb[:, r] * temp[:, c]) * sc for c in range(3)] for r in range(3)])
        g[3:] = _rotation_gradient(x, Rm)
        return f, g


# ---------------------------------------------------------------------------------------------------------------------
# GSL vector_bfgs2 (as ported in pcl/registration/bfgs.h): Fletcher's line search + memoryless BFGS direction update
# ---------------------------------------------------------------------------------------------------------------------
class _Line:
    """phi(alpha) = f(x0 + alpha p) with memoised evaluations (the cost is deterministic, so GSL's last-value caches and a
    full memo return the same numbers)."""

    def __init__(self, cost, x0, f0, g0, p):
        self.cost, self.x0, self.p = cost, x0.copy(), p.copy()
        self.memo = {0.0: (f0, g0.copy())}

    def _eval(self, alpha):
        if alpha not in self.memo:
            self.memo[alpha] = self.cost.fdf(self.x0 + alpha * self.p)
        return self.memo[alpha]

    def f(self, alpha):
        return self._eval(alpha)[0]

    def df(self, alpha):
        return float(np.dot(self._eval(alpha)[1], self.p))

    def x(self, alpha):
        return self.x0 + alpha * self.p

    def g(self, alpha):
        return self._eval(alpha)[1]


def _solve_quadratic(a, b, c):
    if a == 0:
        return [] if b == 0 else [-c / b]
    disc = b * b - 4 * a * c
    if disc > 0:
        if b == 0:
            r = math.sqrt(-c / a)
            return [-r, r]
        temp = -0.5 * (b + (1 if b > 0 else -1) * math.sqrt(disc))
        return sorted([temp / a, c / temp])
    if disc == 0:
        return [-0.5 * b / a] * 2
    return []


def _interp_quad(f0, fp0, f1, zl, zh):
    q = lambda z: f0 + z * (fp0 + z * (f1 - f0 - fp0))
    zmin, fmin = zl, q(zl)
    if q(zh) < fmin:
        zmin, fmin = zh, q(zh)
    c = 2 * (f1 - f0 - fp0)
    if c > 0:
        z = -fp0 / c
        if zl < z < zh and q(z) < fmin:
            zmin = z
    return zmin


def _interp_cubic(f0, fp0, f1, fp1, zl, zh):
    eta = 3 * (f1 - f0) - 2 * fp0 - fp1
    xi = fp0 + fp1 - 2 * (f1 - f0)
    cub = lambda z: f0 + z * (fp0 + z * (eta + z * xi))
    zmin, fmin = zl, cub(zl)
    cands = [zh] + [z for z in _solve_quadratic(3 * xi, 2 * eta, fp0) if zl < z < zh]
    for z in cands:
        if cub(z) < fmin:
            zmin, fmin = z, cub(z)
    return zmin


def _interpolate(a, fa, fpa, b, fb, fpb, xmin, xmax, order=3):
    ymin, ymax = (xmin - a) / (b - a), (xmax - a) / (b - a)
    if ymin > ymax:
        ymin, ymax = ymax, ymin
    if order > 2 and not math.isnan(fpb):
        y = _interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax)
    else:
        y = _interp_quad(fa, fpa * (b - a), fb, ymin, ymax)
    return a + y * (b - a)


def _line_search(line, alpha1, rho=0.01, sigma=0.01, tau1=9.0, tau2=0.05, tau3=0.5):
    """Fletcher's bracketing + sectioning ("Practical Methods of Optimization", as in GSL's linear_minimize.c).
    Returns (status, alpha); status 'ok' or 'noprogress'."""
    f0, fp0 = line.f(0.0), line.df(0.0)
    alpha, alpha_prev = alpha1, 0.0
    falpha_prev, fpalpha_prev = f0, fp0
    a, fa, fpa = 0.0, f0, fp0
    b, fb, fpb = alpha, 0.0, 0.0
    i = 0
    while i < 100:   # bracketing
        i += 1
        falpha = line.f(alpha)
        if falpha > f0 + alpha * rho * fp0 or falpha >= falpha_prev:
            a, fa, fpa = alpha_prev, falpha_prev, fpalpha_prev
            b, fb, fpb = alpha, falpha, math.nan
            break
        fpalpha = line.df(alpha)
        if abs(fpalpha) <= -sigma * fp0:
            return "ok", alpha
        if fpalpha >= 0:
            a, fa, fpa = alpha, falpha, fpalpha
            b, fb, fpb = alpha_prev, falpha_prev, fpalpha_prev
            break
        delta = alpha - alpha_prev
        nxt = _interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta)
        alpha_prev, falpha_prev, fpalpha_prev = alpha, falpha, fpalpha
        alpha = nxt
    while i < 100:   # sectioning (GSL shares the iteration counter between the two loops)
        i += 1
        delta = b - a
        alpha = _interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta)
        falpha = line.f(alpha)
        if (a - alpha) * fpa <= np.finfo(np.float64).eps:
            return "noprogress", alpha
        if falpha > f0 + rho * alpha * fp0 or falpha >= fa:
            b, fb, fpb = alpha, falpha, math.nan
        else:
            fpalpha = line.df(alpha)
            if abs(fpalpha) <= -sigma * fp0:
                return "ok", alpha
            if ((b - a) >= 0 and fpalpha >= 0) or ((b - a) <= 0 and fpalpha <= 0):
                b, fb, fpb = a, fa, fpa
            a, fa, fpa = alpha, falpha, fpalpha
    return "ok", 0.0   # both loops exhausted: GSL reports success and leaves the step at its initial value, 0


def _bfgs_minimize(cost, x):
    """minimizeInit + up to 20 minimizeOneStep / testGradient rounds.  Returns (x, ok)."""
    x = x.copy()
    f, g = cost.fdf(x)
    x0, g0 = x.copy(), g.copy()
    g0norm = float(np.linalg.norm(g0))
    p = g * (-1.0 / g0norm)
    pnorm = float(np.linalg.norm(p))
    fp0 = -g0norm
    delta_f = 0.0
    inner = 0
    while True:
        inner += 1
        if pnorm == 0.0 or g0norm == 0.0 or fp0 == 0.0:
            status = "noprogress"
        else:
            if delta_f < 0:
                dl = max(-delta_f, 10 * np.finfo(np.float64).eps * abs(f))
                alpha1 = min(1.0, 2.0 * dl / (-fp0))
            else:
                alpha1 = 1.0
            line = _Line(cost, x0, f, g0, p)
            status, alpha = _line_search(line, alpha1)
        if status != "ok":
            break
        f_new, g_new, x = line.f(alpha), line.g(alpha).copy(), line.x(alpha)
        delta_f = f_new - f
        f = f_new
        dx0, dg0 = x - x0, g_new - g0
        dxg, dgg, dxdg = float(np.dot(dx0, g_new)), float(np.dot(dg0, g_new)), float(np.dot(dx0, dg0))
        dgnorm = float(np.linalg.norm(dg0))
        A = B = 0.0
        if dxdg != 0:
            B = dxg / dxdg
            A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg
        p = g_new - A * dx0 - B * dg0
        g0, x0 = g_new.copy(), x.copy()
        g0norm, pnorm = float(np.linalg.norm(g0)), float(np.linalg.norm(p))
        direction = -1.0 if float(np.dot(p, g_new)) >= 0.0 else 1.0
        p = p * (direction / pnorm)
        pnorm = float(np.linalg.norm(p))
        fp0 = float(np.dot(p, g0))
        if float(np.linalg.norm(g_new)) < GRAD_TOL:
            status = "success"
            break
        if inner >= MAX_INNER:
            break
    ok = status in ("noprogress", "success") or inner == MAX_INNER
    return x, ok


# ---------------------------------------------------------------------------------------------------------------------
# computeTransformation
# ---------------------------------------------------------------------------------------------------------------------
def gicp_align(src, tgt, max_iterations=10, transformation_epsilon=1e-6, max_correspondence_distance=1.0, guess=None,
               sums="sequential"):
    """Returns dict(T 4x4 float32, converged, iterations, n_corr, evaluations)."""
    src32 = np.ascontiguousarray(src[:, :4], f32)
    tgt32 = np.ascontiguousarray(tgt[:, :4], f32)
    guess = np.eye(4, dtype=f32) if guess is None else np.asarray(guess, f32)
    out = dict(T=np.eye(4, dtype=f32), converged=False, iterations=0, n_corr=0, evaluations=0)
    if tgt32.shape[0] < K_CORR or src32.shape[0] < K_CORR:
        return out
    Ct, Cs = covariances(tgt32), covariances(src32)
    tree = cKDTree(tgt32[:, :3].astype(np.float64))
    transformation = np.eye(4, dtype=f32)
    previous = transformation.copy()
    r2 = max_correspondence_distance ** 2
    nr, converged, n_corr, evals = 0, False, 0, 0
    while not converged:
        TG = np.zeros((4, 4), f32)
        for r in range(4):
            for c in range(4):
                s = f32(0)
                for k in range(4):
                    s = f32(s + transformation[r, k] * guess[k, c])
                TG[r, c] = s
        R = transformation[:3, :].astype(np.float64) @ guess[:, :3].astype(np.float64)
        q = transform_f32(src32, TG)
        _, j = tree.query(q.astype(np.float64))
        d = tgt32[j, :3] - q                                  # float32 differences; d2 with the fma contract
        d64 = d.astype(np.float64)
        d2 = (d64[:, 0] * d64[:, 0]).astype(f32).astype(np.float64)
        d2 = (d64[:, 1] * d64[:, 1] + d2).astype(f32).astype(np.float64)
        d2 = (d64[:, 2] * d64[:, 2] + d2).astype(f32).astype(np.float64)
        keep = np.flatnonzero(d2 < r2)                        # GICP: strict <
        n_corr = int(keep.size)
        previous = transformation.copy()
        if n_corr < 4:
            break
        maha = np.linalg.inv(Ct[j[keep]] + R[None] @ Cs[keep] @ R.T[None])
        cost = _Cost(src32[keep], tgt32[j[keep], :3], maha, guess, sums)
        x, ok = _bfgs_minimize(cost, _state_from_matrix(transformation))
        evals += cost.evaluations
        if not ok:
            break
        transformation = apply_state(np.eye(4, dtype=f32), x)
        ratio = np.full((4, 4), 1.0 / transformation_epsilon)
        ratio[:3, :3] = 1.0 / ROT_EPS
        delta = float((ratio * np.abs(previous.astype(np.float64) - transformation.astype(np.float64))).max())
        nr += 1
        if nr >= max_iterations or delta < 1:
            converged = True
            previous = transformation.copy()
    final = np.eye(4, dtype=f32)
    for r in range(3):
        for c in range(3):
            s = f32(0)
            for k in range(3):
                s = f32(s + previous[r, k] * guess[k, c])
            final[r, c] = s
        final[r, 3] = f32(previous[r, 3] + guess[r, 3])
    out.update(T=final, converged=converged, iterations=nr, n_corr=n_corr, evaluations=evals)
    return out
