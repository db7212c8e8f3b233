"""The quadratic inner objective of GICP (icpslam_amd/csrc/icp_gicp_quadratic.h, icpgpu_params.gicp_inner = QUADRATIC): the
host-side evaluation against a per-point evaluation of the same objective in NumPy.  No GPU: the entry under test,
icpgpu_gicp_quadratic_eval, is pure host arithmetic (the 73 sums a device pass would deliver are built here with NumPy)."""
import ctypes as C

import numpy as np
import pytest

from icpslam_amd import _lib

PAIRS = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (3, 3)]
TRI = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]


def _rot(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _problem(seed, n=4000, spread=60.0):
    rng = np.random.default_rng(seed)
    p = (rng.uniform(-spread, spread, (n, 3))).astype(np.float32)
    base = np.eye(4)
    base[:3, :3] = _rot(*rng.normal(0, 0.02, 3))
    base[:3, 3] = rng.normal(0, 0.3, 3)
    base = base.astype(np.float32)
    q = (p.astype(np.float64) @ base[:3, :3].astype(np.float64).T + base[:3, 3] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    A = rng.normal(0, 1, (n, 3, 3))
    M = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(3)          # symmetric positive definite
    return p, q, M, base


def _sums(p, q, M):
    """the 75 sums of gicp_quadratic_kernel, as (hi, lo) pairs (float64 sums via math.fsum-like pairwise: np.sum of float64)"""
    pt = np.concatenate([p.astype(np.float64), np.ones((len(p), 1))], axis=1)
    qd = q.astype(np.float64)
    Mq = np.einsum("ncd,nd->nc", M, qd)
    out = np.zeros((75, 2))
    for pi, (e, f) in enumerate(PAIRS):
        for k, (c, d) in enumerate(TRI):
            out[pi * 6 + k, 0] = np.sum(pt[:, e] * pt[:, f] * M[:, c, d])
    for e in range(4):
        for c in range(3):
            out[60 + e * 3 + c, 0] = np.sum(pt[:, e] * Mq[:, c])
    out[72, 0] = np.sum(np.einsum("nc,nc->n", qd, Mq))
    out[73, 0] = len(p)
    out[74, 0] = 1.0
    return out


def _state_matrix(base, x):
    """PCL's applyState on the float matrix `base`, in float64 (the library does it in float32: compared loosely below)"""
    T = base.astype(np.float64).copy()
    T[:3, :3] = _rot(x[3], x[4], x[5]) @ T[:3, :3]
    T[:3, 3] += x[:3]
    return T


def _direct(p, q, M, base, T):
    """f and the raw sums of the smooth objective, per point, float64"""
    pd, qd = p.astype(np.float64), q.astype(np.float64)
    r = pd @ T[:3, :3].T + T[:3, 3] - qd
    t = np.einsum("ncd,nd->nc", M, r)
    b = pd @ base[:3, :3].astype(np.float64).T + base[:3, 3].astype(np.float64)
    return np.sum(np.einsum("nc,nc->n", r, t)), t.sum(0), np.einsum("na,nc->ac", b, t)


@pytest.fixture(scope="module")
def lib():
    try:
        return _lib.load()
    except ImportError as e:  # pragma: no cover
        pytest.skip(str(e))


@pytest.mark.parametrize("seed", range(6))
def test_quadratic_eval_matches_per_point_evaluation(lib, seed):
    p, q, M, base = _problem(seed)
    sums = np.ascontiguousarray(_sums(p, q, M).reshape(-1))
    rng = np.random.default_rng(100 + seed)
    dp = C.POINTER(C.c_double)
    base_cm = np.ascontiguousarray(base.T.reshape(-1).astype(np.float32))      # column-major
    for trial in range(8):
        x = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.01, 3)]) * (0.0 if trial == 0 else 1.0)
        f, g = C.c_double(), np.zeros(6)
        rc = lib.icpgpu_gicp_quadratic_eval(sums.ctypes.data_as(dp), base_cm.ctypes.data_as(C.POINTER(C.c_float)),
                                            np.ascontiguousarray(x).ctypes.data_as(dp), C.byref(f), g.ctypes.data_as(dp))
        assert rc == 0
        # the float matrix the library forms (float32 quaternion products) differs from the float64 one by ~1e-7 per entry: use a
        # finite-difference-free check -- evaluate the reference at the float64 matrix and allow for that difference
        T = _state_matrix(base, x)
        s1, s2, s5 = _direct(p, q, M, base, T)
        m = float(len(p))
        f_ref = s1 / m
        # sensitivity of f to a 1e-7 perturbation of T's entries: |df| <= |grad_T f| * 1e-7 ~ 2/m * |sum p~ (M r)| * 1e-7
        pt = np.concatenate([p.astype(np.float64), np.ones((len(p), 1))], axis=1)
        r = p.astype(np.float64) @ T[:3, :3].T + T[:3, 3] - q.astype(np.float64)
        t = np.einsum("ncd,nd->nc", M, r)
        slack = 2.0 / m * np.abs(np.einsum("ne,nc->ec", pt, t)).sum() * 3e-7 + 1e-12 * abs(f_ref)
        assert abs(f.value - f_ref) <= slack, (f.value, f_ref, slack)
        # translation gradient: 2/m sum M r
        g_t = 2.0 / m * s2
        slack_g = 2.0 / m * np.abs(np.einsum("ncd,ne->cd", M, np.abs(pt))).sum() * 3e-7 + 1e-9 * np.abs(g_t).max()
        assert np.all(np.abs(g[:3] - g_t) <= slack_g), (g[:3], g_t, slack_g)


def test_quadratic_eval_is_exact_for_a_representable_matrix(lib):
    """x = 0: the state matrix is `base` itself, no float32 quaternion arithmetic in the way -- the quadratic form must then agree
    with the per-point evaluation to the rounding of the NumPy sums (1e-10 relative here; the device delivers exact sums)."""
    p, q, M, base = _problem(42, n=2000, spread=20.0)
    sums = np.ascontiguousarray(_sums(p, q, M).reshape(-1))
    dp = C.POINTER(C.c_double)
    base_cm = np.ascontiguousarray(base.T.reshape(-1).astype(np.float32))
    x = np.zeros(6)
    f, g = C.c_double(), np.zeros(6)
    assert lib.icpgpu_gicp_quadratic_eval(sums.ctypes.data_as(dp), base_cm.ctypes.data_as(C.POINTER(C.c_float)), x.ctypes.data_as(dp),
                                          C.byref(f), g.ctypes.data_as(dp)) == 0
    T = base.astype(np.float64)
    s1, s2, s5 = _direct(p, q, M, base, T)
    m = float(len(p))
    # cancellation: the sums are ~ m |p|^2 |M|, f ~ m |r|^2 |M| -- NumPy's float64 sums carry ~1e-16 * sqrt(m) of the former
    big = m * (20.0 ** 2) * 3.0
    assert abs(f.value * m - s1) <= 1e-12 * big
    assert np.all(np.abs(g[:3] * m / 2.0 - s2) <= 1e-12 * big)
    # rotation gradient at x = 0: g[3..5] = 2/m * matricesInnerProd(dR/dangle, sum (base p)(M r)^T)
    dRoll = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], float)
    dPitch = np.array([[0, 0, 1], [0, 0, 0], [-1, 0, 0]], float)
    dYaw = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 0]], float)
    Rm = 2.0 / m * s5
    ref = [np.sum(D.T * Rm) for D in (dRoll, dPitch, dYaw)]   # PCL matricesInnerProd: sum_ij D(j,i) R(i,j)
    assert np.allclose(g[3:], ref, rtol=0, atol=1e-12 * big * 20.0 / m * 2)
