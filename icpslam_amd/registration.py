"""Host-side mirror of the PCL `Registration` protocol the reference drives, on top of the C-ABI.

The reference calls (identically at /root/reference/src/icpslam/icp_odometer.cpp:188-201 and
src/icpslam/octree_mapper.cpp:104-117):

    icp.setMaximumIterations(ICP_MAX_ITERS); icp.setTransformationEpsilon(ICP_EPSILON);
    icp.setMaxCorrespondenceDistance(ICP_MAX_CORR_DIST); icp.setRANSACIterations(0);
    icp.setInputSource(curr); icp.setInputTarget(prev); icp.align(out);
    T = icp.getFinalTransformation(); icp.hasConverged(); icp.getFitnessScore()

`IterativeClosestPoint` below keeps those method names, argument meaning and error behaviour
(failure = hasConverged() False, never an exception on the data path) so the parity tests read like
a PCL caller.  `Context` is the thin 1:1 wrapper over include/icpgpu.h used by bench.py.
The C++ twin of this class is include/icpgpu_registration.hpp.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import IcpGpuError, Params, Profile, Result


def _as_cloud(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("cloud must be (N, 4) float32: the pcl::PointXYZ layout x, y, z, pad")
    return a


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _colmajor16(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).reshape(4, 4).T).reshape(16)


class Context:
    """One icpgpu_ctx: one device, one HIP stream, reusable scratch."""

    def __init__(self, device_id: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.icpgpu_create(C.byref(h), int(device_id))
        if rc != 0:
            raise IcpGpuError(rc, self._L.icpgpu_last_error(None).decode())
        self._h = h
        self.n_source = 0
        self.n_target = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.icpgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise IcpGpuError(rc, self._L.icpgpu_last_error(self._h).decode())

    # parameters -----------------------------------------------------------------------------------------
    def default_params(self) -> Params:
        p = Params()
        self._L.icpgpu_default_params(C.byref(p))
        return p

    def get_params(self) -> Params:
        p = Params()
        self._check(self._L.icpgpu_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, p: Params | None = None, **kw):
        p = p or self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        self._check(self._L.icpgpu_set_params(self._h, C.byref(p)))

    # inputs ----------------------------------------------------------------------------------------------
    def set_source(self, cloud):
        cloud = _as_cloud(cloud)
        self._check(self._L.icpgpu_set_source(self._h, _fp(cloud), cloud.shape[0]))
        self.n_source = cloud.shape[0]

    def set_target(self, cloud):
        cloud = _as_cloud(cloud)
        self._check(self._L.icpgpu_set_target(self._h, _fp(cloud), cloud.shape[0]))
        # the library may have recognised the context's current SOURCE in `cloud` and taken the promote path (icpgpu.h)
        ns, nt = C.c_size_t(), C.c_size_t()
        self._check(self._L.icpgpu_cloud_sizes(self._h, C.byref(ns), C.byref(nt)))
        self.n_source, self.n_target = int(ns.value), int(nt.value)

    def set_source_device(self, ptr: int, n: int):
        self._check(self._L.icpgpu_set_source_device(self._h, C.c_void_p(ptr), n))
        self.n_source = n

    def set_target_device(self, ptr: int, n: int):
        self._check(self._L.icpgpu_set_target_device(self._h, C.c_void_p(ptr), n))
        self.n_target = n

    def promote_source_to_target(self):
        self._check(self._L.icpgpu_promote_source_to_target(self._h))
        self.n_target, self.n_source = self.n_source, 0

    # the mapper's map (octree_mapper.cpp:55-90) -----------------------------------------------
    def map_set_search(self, pcl_approx: bool):
        """False: exact nearest map point (default); True: PCL's approxNearestSearch as octree_mapper.cpp:84 calls it."""
        self._check(self._L.icpgpu_map_set_search(self._h, 1 if pcl_approx else 0))

    def map_reset(self, resolution: float = 0.5):
        """resetMap(): empty one-point-per-voxel map (octree_resolution_, octree_mapper.cpp:41)."""
        self._check(self._L.icpgpu_map_reset(self._h, float(resolution)))

    def map_add_points(self, cloud, pose=None) -> int:
        """addPointsToMap(transformCloudToPoseFrame(cloud, pose)); returns the number of points appended."""
        cloud = _as_cloud(cloud)
        n = C.c_size_t()
        g = _fp(_colmajor16(pose)) if pose is not None else None
        self._check(self._L.icpgpu_map_add_points(self._h, _fp(cloud), cloud.shape[0], g, C.byref(n)))
        return int(n.value)

    def map_add_source(self, pose=None) -> int:
        n = C.c_size_t()
        g = _fp(_colmajor16(pose)) if pose is not None else None
        self._check(self._L.icpgpu_map_add_source(self._h, g, C.byref(n)))
        return int(n.value)

    def map_size(self) -> int:
        n = C.c_size_t()
        self._check(self._L.icpgpu_map_size(self._h, C.byref(n)))
        return int(n.value)

    def map_points(self) -> np.ndarray:
        n = self.map_size()
        out = np.empty((n, 4), np.float32)
        m = C.c_size_t()
        self._check(self._L.icpgpu_map_get_points(self._h, _fp(out) if n else None, n, C.byref(m)))
        return out

    def map_nn_target(self, pose, pose_inv, want_cloud: bool = True):
        """approxNearestNeighbors(cloud_in_map) moved by pose_inv becomes the target (exact NN); returns the nn cloud."""
        out = np.empty((self.n_source, 4), np.float32) if want_cloud else None
        n = C.c_size_t()
        self._check(self._L.icpgpu_map_nn_target(self._h, _fp(_colmajor16(pose)), _fp(_colmajor16(pose_inv)),
                                                 _fp(out) if out is not None and self.n_source else None, C.byref(n)))
        self.n_target = int(n.value)
        return out[: n.value] if out is not None else None

    # hot path ---------------------------------------------------------------------------------------------
    def align(self, guess=None, want_cloud: bool = False, want_fitness: bool = False):
        res = Result()
        g = None
        if guess is not None:
            gbuf = _colmajor16(guess)
            g = _fp(gbuf)
        out = np.empty((self.n_source, 4), np.float32) if want_cloud else None
        self._check(self._L.icpgpu_align(self._h, g, _fp(out) if out is not None else None, int(want_fitness),
                                         C.byref(res)))
        return result_dict(res, out)

    def align_view(self, guess=None, want_fitness: bool = False):
        """icpgpu_align_view: align with the aligned cloud taken from the context's staging buffer (copied here: the view lives until
        the context's next call)."""
        res = Result()
        g = None
        if guess is not None:
            gbuf = _colmajor16(guess)
            g = _fp(gbuf)
        view = C.POINTER(C.c_float)()
        n_out = C.c_size_t()
        self._check(self._L.icpgpu_align_view(self._h, g, int(want_fitness), C.byref(res), C.byref(view), C.byref(n_out)))
        out = np.ctypeslib.as_array(view, shape=(n_out.value, 4)).copy() if n_out.value else np.empty((0, 4), np.float32)
        return result_dict(res, out)

    def fitness(self, max_range: float = float(np.finfo(np.float64).max)) -> float:
        v = C.c_double()
        self._check(self._L.icpgpu_fitness(self._h, float(max_range), C.byref(v)))
        return v.value

    def align_batch(self, sources, targets, want_fitness: bool = False):
        n = len(sources)
        srcs = [_as_cloud(s) for s in sources]
        tgts = [_as_cloud(t) for t in targets]
        FP = C.POINTER(C.c_float)
        sp = (FP * n)(*[_fp(s) for s in srcs])
        tp = (FP * n)(*[_fp(t) for t in tgts])
        ns = (C.c_size_t * n)(*[s.shape[0] for s in srcs])
        nt = (C.c_size_t * n)(*[t.shape[0] for t in tgts])
        res = (Result * n)()
        self._check(self._L.icpgpu_align_batch(self._h, n, sp, ns, tp, nt, int(want_fitness), res))
        return [result_dict(r, None) for r in res]

    # kernel-level entry points ------------------------------------------------------------------------------
    def nn(self, T=np.eye(4)):
        idx = np.empty(self.n_source, np.int32)
        d2 = np.empty(self.n_source, np.float32)
        Tb = _colmajor16(T)
        self._check(self._L.icpgpu_nn(self._h, _fp(Tb), idx.ctypes.data_as(C.POINTER(C.c_int32)), _fp(d2)))
        return idx, d2

    def reduce(self, T, max_dist: float) -> np.ndarray:
        sums = np.zeros(17, np.float64)
        Tb = _colmajor16(T)
        self._check(self._L.icpgpu_reduce(self._h, _fp(Tb), float(max_dist), sums.ctypes.data_as(C.POINTER(C.c_double))))
        return sums

    def solve(self, sums) -> np.ndarray:
        sums = np.ascontiguousarray(sums, np.float64)
        Tk = np.zeros(16, np.float64)
        dp = C.POINTER(C.c_double)
        rc = self._L.icpgpu_solve(sums.ctypes.data_as(dp), Tk.ctypes.data_as(dp))
        if rc != 0:
            raise IcpGpuError(rc, "solve failed (n < 1 or non-finite sums)")
        return Tk.reshape(4, 4).T.copy()

    def transform(self, T) -> np.ndarray:
        out = np.empty((self.n_source, 4), np.float32)
        Tb = _colmajor16(T)
        self._check(self._L.icpgpu_transform(self._h, _fp(Tb), _fp(out)))
        return out

    def gicp_covariances(self, of_target: bool = False) -> np.ndarray:
        """(n, 3, 3) regularised neighbourhood covariances of the source (or target) cloud -- GICP row a11."""
        n = self.n_target if of_target else self.n_source
        out = np.zeros((n, 6), np.float64)
        self._check(self._L.icpgpu_gicp_covariances(self._h, int(of_target), out.ctypes.data_as(C.POINTER(C.c_double))))
        full = np.empty((n, 3, 3))
        full[:, 0, 0], full[:, 0, 1], full[:, 0, 2] = out[:, 0], out[:, 1], out[:, 2]
        full[:, 1, 0], full[:, 1, 1], full[:, 1, 2] = out[:, 1], out[:, 3], out[:, 4]
        full[:, 2, 0], full[:, 2, 1], full[:, 2, 2] = out[:, 2], out[:, 4], out[:, 5]
        return full

    def gicp_quadratic_sums(self, T=None) -> np.ndarray:
        """(75, 2) the sums of GICP's quadratic inner objective at transform T as (hi, lo) pairs (icp_gicp_quadratic.h) -- the
        device half of params.gicp_inner = GICP_INNER_QUADRATIC, for tests."""
        out = np.zeros((75, 2), np.float64)
        Tb = None if T is None else _colmajor16(T)
        self._check(self._L.icpgpu_gicp_quadratic_sums(self._h, None if Tb is None else _fp(Tb), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    # the step before the path (icp_odometer.cpp:96-101) ---------------------------------------------------------
    def voxel_grid(self, cloud, leaf: float) -> np.ndarray:
        cloud = _as_cloud(cloud)
        out = np.empty_like(cloud)
        n_out = C.c_size_t()
        self._check(self._L.icpgpu_voxel_grid(self._h, _fp(cloud), cloud.shape[0], float(leaf), _fp(out), C.byref(n_out)))
        return out[: n_out.value].copy()

    def voxel_grid_view(self, cloud, leaf: float) -> np.ndarray:
        """icpgpu_voxel_grid_view: the filtered cloud copied out of the context's staging buffer (one host copy; the view itself is
        only valid until the context's next call)."""
        cloud = _as_cloud(cloud)
        view = C.POINTER(C.c_float)()
        n_out = C.c_size_t()
        self._check(self._L.icpgpu_voxel_grid_view(self._h, _fp(cloud), cloud.shape[0], float(leaf), C.byref(view), C.byref(n_out)))
        if n_out.value == 0:
            return np.empty((0, 4), np.float32)
        return np.ctypeslib.as_array(view, shape=(n_out.value, 4)).copy()

    def set_source_voxel_filtered(self, cloud, leaf: float) -> int:
        cloud = _as_cloud(cloud)
        n_out = C.c_size_t()
        self._check(self._L.icpgpu_set_source_voxel_filtered(self._h, _fp(cloud), cloud.shape[0], float(leaf), C.byref(n_out)))
        self.n_source = int(n_out.value)
        return self.n_source

    # measurement -----------------------------------------------------------------------------------------------
    def calibrate(self) -> int:
        """icpgpu_calibrate: time GICP's two inner solvers on the clouds this context holds and keep the faster (GICP_SOLVER_*)."""
        v = C.c_int()
        self._check(self._L.icpgpu_calibrate(self._h, C.byref(v)))
        return int(v.value)

    def profile_reset(self):
        self._check(self._L.icpgpu_profile_reset(self._h))

    def profile_sampling(self, every: int):
        """Time one correspondence sweep in `every` (library default 7; 1 = all of them, at 6-7 us per iteration)."""
        self._check(self._L.icpgpu_profile_set_sampling(self._h, int(every)))

    def profile(self) -> Profile:
        p = Profile()
        self._check(self._L.icpgpu_profile_get(self._h, C.byref(p)))
        return p

    def synchronize(self):
        self._check(self._L.icpgpu_synchronize(self._h))

    def count_candidates(self, enable: bool = True):
        """Counting runs: target points evaluated by the grid sweeps (see icpgpu_count_candidates)."""
        self._check(self._L.icpgpu_count_candidates(self._h, int(enable)))

    def candidates(self) -> int:
        v = C.c_uint64()
        self._check(self._L.icpgpu_count_candidates_read(self._h, C.byref(v)))
        return int(v.value)


def result_dict(res: Result, cloud):
    return dict(T=np.array(res.T, dtype=np.float32).reshape(4, 4).T.copy(), converged=bool(res.converged),
                iterations=int(res.iterations), state=int(res.convergence_state), n_corr=int(res.n_correspondences),
                mse=float(res.mse_last), fitness=float(res.fitness), cloud=cloud, t_total_ms=float(res.t_total_ms),
                t_device_ms=float(res.t_device_ms), gicp_solver=int(res.gicp_solver))


class IterativeClosestPoint:
    """pcl::IterativeClosestPoint<PointXYZ, PointXYZ>-shaped front end (same method names as the reference uses):
    point-to-point ICP, the solver BASELINE.json's north_star specifies.  The class the reference literally instantiates
    is GeneralizedIterativeClosestPoint below; each mirror keeps the semantics of the PCL class it is named after."""

    _shared_ctx: dict = {}
    METHOD = _lib.P2P_SVD

    def __init__(self, device_id: int = 0, method: int | None = None):
        method = self.METHOD if method is None else method
        # the reference builds a fresh registration object per scan (icp_odometer.cpp:188); the GPU context is
        # cached per device so that doing the same here costs nothing
        ctx = IterativeClosestPoint._shared_ctx.get(device_id)
        if ctx is None or ctx._h is None:
            ctx = Context(device_id)
            IterativeClosestPoint._shared_ctx[device_id] = ctx
        self._ctx = ctx
        self._params = ctx.default_params()
        self._params.method = method
        self._source = None
        self._target = None
        self._result = None
        self._fitness = None

    # setters used by the reference -----------------------------------------------------------------------------
    def setMaximumIterations(self, n):           # icp_odometer.cpp:189 (passes a double constant)
        self._params.max_iterations = int(n)

    def setTransformationEpsilon(self, eps):     # icp_odometer.cpp:190
        self._params.transformation_epsilon = float(eps)

    def setMaxCorrespondenceDistance(self, d):   # icp_odometer.cpp:191
        self._params.max_correspondence_distance = float(d)

    def setRANSACIterations(self, n):            # icp_odometer.cpp:192 -- always 0 in the reference
        if int(n) != 0:
            raise NotImplementedError("RANSAC outlier rejection is not on the reference's path (always 0)")

    def setEuclideanFitnessEpsilon(self, eps):
        self._params.euclidean_fitness_epsilon = float(eps)

    def setInputSource(self, cloud):             # icp_odometer.cpp:193
        self._source = _as_cloud(cloud)

    def setInputTarget(self, cloud):             # icp_odometer.cpp:194
        self._target = _as_cloud(cloud)

    # the call ------------------------------------------------------------------------------------------------------
    def align(self, guess=None) -> np.ndarray:   # icp_odometer.cpp:198; returns the aligned source cloud
        if self._source is None or self._target is None:
            raise IcpGpuError(_lib.ERR_NO_INPUT, "align: setInputSource/setInputTarget first")
        self._ctx.set_params(self._params)
        self._ctx.set_source(self._source)
        self._ctx.set_target(self._target)
        self._result = self._ctx.align(guess=guess, want_cloud=True)
        self._ctx._last_user = self          # objects of one device share the cached context (see getFitnessScore)
        return self._result["cloud"]

    def getFinalTransformation(self) -> np.ndarray:   # icp_odometer.cpp:199
        return np.eye(4, dtype=np.float32) if self._result is None else self._result["T"]

    def hasConverged(self) -> bool:                   # icp_odometer.cpp:201
        return bool(self._result and self._result["converged"])

    def getFitnessScore(self, max_range: float = float(np.finfo(np.float64).max)) -> float:   # icp_odometer.cpp:201
        if self._result is None:
            raise IcpGpuError(_lib.ERR_NO_INPUT, "getFitnessScore before align")
        if getattr(self._ctx, "_last_user", None) is not self:
            # another registration object used the shared context since this one's align: put this object's clouds back
            # and evaluate under ITS transform (kernel-level entry points), not under whatever the context did last
            self._ctx.set_params(self._params)
            self._ctx.set_source(self._source)
            self._ctx.set_target(self._target)
            self._ctx.nn(self._result["T"])
            s = self._ctx.reduce(self._result["T"], 1e18 if max_range >= 1e36 else float(np.sqrt(max_range)))
            self._ctx._last_user = None
            return float(s[16] / s[0]) if s[0] > 0 else float(np.finfo(np.float64).max)
        return self._ctx.fitness(max_range)

    @property
    def result(self):
        return self._result


class GeneralizedIterativeClosestPoint(IterativeClosestPoint):
    """pcl::GeneralizedIterativeClosestPoint<PointXYZ, PointXYZ>-shaped front end: the class the reference instantiates at
    icp_odometer.cpp:188 and octree_mapper.cpp:104 (plane-to-plane cost, BFGS inner solver, PCL's constructor defaults)."""

    METHOD = _lib.GICP

    def setQuadraticInnerSolver(self, on: bool):
        """NOT a PCL method (the C++ shim has the same one): the inner minimisation on the quadratic form of each outer iteration,
        icpgpu_params.gicp_inner -- faster, within tolerance of the default's result instead of on its bits (include/icpgpu.h)."""
        self._params.gicp_inner = _lib.GICP_INNER_QUADRATIC if on else _lib.GICP_INNER_EXACT
