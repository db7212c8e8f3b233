"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/icp_oracle.h).  May be imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by icpslam_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

P2P_SVD, GICP = 0, 1
NN_KDTREE, NN_BRUTE = 0, 1
PREC_F64, PREC_PCL_F32 = 0, 1
ARITH_FMA, ARITH_FLANN = 0, 1
GICP_SUMS_EXACT, GICP_SUMS_SEQUENTIAL, GICP_SUMS_SEQUENTIAL_REVERSED, GICP_SUMS_SMOOTH = 0, 1, 2, 3
STATE_NAMES = {0: "NOT_CONVERGED", 1: "ITERATIONS", 2: "TRANSFORM", 3: "ABS_MSE", 4: "REL_MSE",
               5: "NO_CORRESPONDENCES"}


class Params(C.Structure):
    _fields_ = [("method", C.c_int), ("max_iterations", C.c_int), ("transformation_epsilon", C.c_double),
                ("max_correspondence_distance", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
                ("min_correspondences", C.c_int), ("force_iterations", C.c_int), ("nn_mode", C.c_int),
                ("precision", C.c_int), ("arith", C.c_int), ("gicp_sums", C.c_int)]


class Result(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("converged", C.c_int), ("iterations", C.c_int),
                ("convergence_state", C.c_int), ("n_correspondences", C.c_uint), ("mse_last", C.c_double),
                ("fitness", C.c_double)]

    def matrix(self) -> np.ndarray:
        return np.array(self.T, dtype=np.float32).reshape(4, 4).T.copy()   # column-major -> numpy


class IterTrace(C.Structure):
    _fields_ = [("Tk", C.c_double * 16), ("final", C.c_double * 16), ("sums", C.c_double * 17),
                ("n_corr", C.c_uint), ("mse", C.c_double)]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, f) for f in ("icp_oracle.c", "gicp_oracle.c", "map_oracle.c", "icp_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_nn.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.c_int, C.c_int, ip, fp]
        L.orc_reduce.argtypes = [fp, C.c_size_t, fp, fp, ip, fp, C.c_double, dp]
        L.orc_umeyama.argtypes = [dp, dp]
        L.orc_transform_cloud.argtypes = [fp, C.c_size_t, fp, fp]
        L.orc_transform_cloud.restype = None
        L.orc_fitness.argtypes = [fp, C.c_size_t, fp, C.c_size_t, fp, C.c_double, C.c_int, C.c_int]
        L.orc_fitness.restype = C.c_double
        L.orc_icp_align.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.POINTER(Params), fp, fp, C.c_int,
                                    C.POINTER(Result), C.POINTER(IterTrace)]
        L.orc_voxel_grid.argtypes = [fp, C.c_size_t, C.c_float, fp]
        L.orc_voxel_grid.restype = C.c_long
        L.orc_gicp_covariances.argtypes = [fp, C.c_size_t, C.c_int, dp]
        L.orc_gicp_covariances_ex.argtypes = [fp, C.c_size_t, C.c_int, C.c_int, dp]
        L.orc_svd3.argtypes = [dp, dp, dp, dp]
        L.orc_svd3.restype = None
        L.orc_svd3_eigen_u.argtypes = [dp, dp, dp]
        L.orc_svd3_eigen_u.restype = None
        L.orc_map_create.argtypes = [C.c_double]
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_destroy.restype = None
        L.orc_map_size.argtypes = [C.c_void_p]
        L.orc_map_size.restype = C.c_size_t
        L.orc_map_points.argtypes = [C.c_void_p]
        L.orc_map_points.restype = fp
        L.orc_map_add_points.argtypes = [C.c_void_p, fp, C.c_size_t, fp]
        L.orc_map_add_points.restype = C.c_long
        L.orc_map_add_points_sequential.argtypes = [C.c_void_p, fp, C.c_size_t, fp]
        L.orc_map_add_points_sequential.restype = C.c_long
        L.orc_map_nn_cloud.argtypes = [C.c_void_p, fp, C.c_size_t, fp, fp, fp]
        L.orc_map_nn_cloud.restype = C.c_long
        _lib = L
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _colmajor(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def nn(src, tgt, T=np.eye(4), nn_mode=NN_KDTREE, arith=ARITH_FMA):
    src, ps = _f32(src)
    tgt, pt = _f32(tgt)
    Tc, pT = _f32(_colmajor(T))
    idx = np.empty(src.shape[0], np.int32)
    d2 = np.empty(src.shape[0], np.float32)
    lib().orc_nn(ps, src.shape[0], pt, tgt.shape[0], pT, nn_mode, arith,
                 idx.ctypes.data_as(C.POINTER(C.c_int32)), d2.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, d2


def reduce(src, tgt, T, idx, d2, max_corr_dist):
    src, ps = _f32(src)
    tgt, pt = _f32(tgt)
    Tc, pT = _f32(_colmajor(T))
    idx = np.ascontiguousarray(idx, np.int32)
    d2 = np.ascontiguousarray(d2, np.float32)
    sums = np.zeros(17, np.float64)
    lib().orc_reduce(ps, src.shape[0], pt, pT, idx.ctypes.data_as(C.POINTER(C.c_int32)),
                     d2.ctypes.data_as(C.POINTER(C.c_float)), float(max_corr_dist),
                     sums.ctypes.data_as(C.POINTER(C.c_double)))
    return sums


def umeyama(sums) -> np.ndarray:
    sums = np.ascontiguousarray(sums, np.float64)
    Tk = np.zeros(16, np.float64)
    lib().orc_umeyama(sums.ctypes.data_as(C.POINTER(C.c_double)), Tk.ctypes.data_as(C.POINTER(C.c_double)))
    return Tk.reshape(4, 4).T.copy()


def transform_cloud(cloud, T):
    cloud, pc = _f32(cloud)
    Tc, pT = _f32(_colmajor(T))
    out = np.empty_like(cloud)
    lib().orc_transform_cloud(pc, cloud.shape[0], pT, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def fitness(src, tgt, T, max_range=float(np.finfo(np.float64).max), nn_mode=NN_KDTREE, arith=ARITH_FMA):
    src, ps = _f32(src)
    tgt, pt = _f32(tgt)
    Tc, pT = _f32(_colmajor(T))
    return lib().orc_fitness(ps, src.shape[0], pt, tgt.shape[0], pT, float(max_range), nn_mode, arith)


def icp_align(src, tgt, params: Params | None = None, guess=None, want_cloud=False, want_fitness=False,
              want_trace=False):
    """Returns dict(T (4x4 float32), converged, iterations, state, n_corr, mse, fitness, cloud, trace)."""
    params = params or default_params()
    src, ps = _f32(src)
    tgt, pt = _f32(tgt)
    res = Result()
    out = np.empty_like(src) if want_cloud else None
    trace = (IterTrace * max(1, params.max_iterations))() if want_trace else None
    g = None
    if guess is not None:
        gc, g = _f32(_colmajor(guess))
    rc = lib().orc_icp_align(ps, src.shape[0], pt, tgt.shape[0], C.byref(params), g,
                             out.ctypes.data_as(C.POINTER(C.c_float)) if out is not None else None,
                             int(want_fitness), C.byref(res), trace)
    if rc != 0:
        raise RuntimeError(f"orc_icp_align rc={rc}")
    tr = None
    if want_trace:
        tr = [dict(Tk=np.array(t.Tk).reshape(4, 4).T.copy(), final=np.array(t.final).reshape(4, 4).T.copy(),
                   sums=np.array(t.sums), n_corr=int(t.n_corr), mse=float(t.mse))
              for t in trace[: res.iterations]]
    return dict(T=res.matrix(), converged=bool(res.converged), iterations=int(res.iterations),
                state=int(res.convergence_state), n_corr=int(res.n_correspondences), mse=float(res.mse_last),
                fitness=float(res.fitness), cloud=out, trace=tr)


def voxel_grid(cloud, leaf: float):
    cloud, pc = _f32(cloud)
    out = np.empty_like(cloud)
    n = lib().orc_voxel_grid(pc, cloud.shape[0], float(leaf), out.ctypes.data_as(C.POINTER(C.c_float)))
    if n < 0:
        return cloud.copy()
    return out[:n].copy()


def gicp_covariances(cloud, arith=ARITH_FMA, pcl_order=False) -> np.ndarray:
    cloud, pc = _f32(cloud)
    out = np.zeros((cloud.shape[0], 9), np.float64)
    rc = lib().orc_gicp_covariances_ex(pc, cloud.shape[0], arith, int(pcl_order), out.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise RuntimeError("orc_gicp_covariances: cloud smaller than k = 20")
    return out.reshape(-1, 3, 3)


def svd3_eigen_u(A):
    """Eigen::JacobiSVD<Matrix3d>(A, ComputeFullU): (U, singular values) as gicp_oracle.c restates it (computeCovariances)."""
    A = np.ascontiguousarray(A, np.float64)
    U, s = np.empty((3, 3)), np.empty(3)
    dp = C.POINTER(C.c_double)
    lib().orc_svd3_eigen_u(A.ctypes.data_as(dp), U.ctypes.data_as(dp), s.ctypes.data_as(dp))
    return U, s


def svd3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    U, s, V = np.zeros(9), np.zeros(3), np.zeros(9)
    dp = C.POINTER(C.c_double)
    lib().orc_svd3(A.ctypes.data_as(dp), U.ctypes.data_as(dp), s.ctypes.data_as(dp), V.ctypes.data_as(dp))
    return U.reshape(3, 3), s, V.reshape(3, 3)


class VoxelMap:
    """The mapper's one-point-per-voxel map (oracle/map_oracle.c; octree_mapper.cpp:55-90), sequential restatement."""

    def __init__(self, resolution: float = 0.5):
        self._L = lib()
        self._h = self._L.orc_map_create(float(resolution))
        if not self._h:
            raise MemoryError("orc_map_create")

    def close(self):
        if getattr(self, "_h", None):
            self._L.orc_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._L.orc_map_size(self._h))

    def add_points(self, cloud, pose=None, sequential: bool = False) -> int:
        """sequential=True: the reference's loop as written (O(map) per insertion); default: the batch form of the same rule."""
        cloud, pc = _f32(cloud)
        g = None
        if pose is not None:
            gc, g = _f32(_colmajor(pose))
        add = self._L.orc_map_add_points_sequential if sequential else self._L.orc_map_add_points
        n = add(self._h, pc, cloud.shape[0], g)
        if n < 0:
            raise MemoryError("orc_map_add_points")
        return int(n)

    def points(self) -> np.ndarray:
        n = len(self)
        if n == 0:
            return np.zeros((0, 4), np.float32)
        return np.ctypeslib.as_array(self._L.orc_map_points(self._h), shape=(n, 4)).copy()

    def nn_cloud(self, cloud, pose, pose_inv) -> np.ndarray:
        cloud, pc = _f32(cloud)
        a, pa = _f32(_colmajor(pose))
        b, pb = _f32(_colmajor(pose_inv))
        out = np.empty_like(cloud)
        n = self._L.orc_map_nn_cloud(self._h, pc, cloud.shape[0], pa, pb, out.ctypes.data_as(C.POINTER(C.c_float)))
        if n < 0:
            raise MemoryError("orc_map_nn_cloud")
        return out[:n].copy()
