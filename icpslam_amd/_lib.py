"""ctypes binding of include/icpgpu.h (libicpgpu.so).  Fails loudly when the HIP library is missing:
there is no CPU or PyTorch fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICPGPU_FLAVOUR=dev: the development flavour (libicpgpu_dev.so, built with -DICPGPU_DEV_SWITCHES: kernel variants, tuning
# constants and test modes behind environment switches -- icpslam_amd/csrc/icp_env.h).  A process loads ONE flavour; the
# tests of the development modes run in sub-processes (tests/conftest.py: the dev_flavour fixture).  Default: the release library.
FLAVOUR = "dev" if os.environ.get("ICPGPU_FLAVOUR", "") == "dev" else "release"
LIB_PATH = os.path.join(_HERE, "libicpgpu_dev.so" if FLAVOUR == "dev" else "libicpgpu.so")
if os.environ.get("ICPGPU_LIB_PATH"):   # A/B builds of an experiment (scripts/): an explicit library file
    LIB_PATH = os.environ["ICPGPU_LIB_PATH"]

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_NO_INPUT, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
P2P_SVD, GICP = 0, 1
GICP_INNER_EXACT, GICP_INNER_QUADRATIC = 0, 1
GICP_SOLVER_NONE, GICP_SOLVER_HOST, GICP_SOLVER_DEVICE, GICP_SOLVER_QUADRATIC = 0, 1, 2, 3
HEADER_VERSION = 1001          # the icpgpu.h these mirrors were written against (ICPGPU_HEADER_VERSION)
NN_AUTO, NN_BRUTE, NN_GRID = 0, 1, 2
STATE_NAMES = {0: "NOT_CONVERGED", 1: "ITERATIONS", 2: "TRANSFORM", 3: "ABS_MSE", 4: "REL_MSE",
               5: "NO_CORRESPONDENCES"}


class Params(C.Structure):
    _fields_ = [("method", C.c_int32), ("max_iterations", C.c_int32), ("transformation_epsilon", C.c_double),
                ("max_correspondence_distance", C.c_double), ("euclidean_fitness_epsilon", C.c_double),
                ("min_correspondences", C.c_int32), ("force_iterations", C.c_int32), ("nn_mode", C.c_int32),
                ("brute_variant", C.c_int32), ("gicp_inner", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32),
                ("convergence_state", C.c_int32), ("n_correspondences", C.c_uint32), ("mse_last", C.c_double),
                ("fitness", C.c_double), ("t_total_ms", C.c_double), ("t_device_ms", C.c_double),
                ("gicp_solver", C.c_int32), ("reserved0", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("nn_launches", C.c_uint64), ("nn_ms", C.c_double), ("nn_pairs", C.c_uint64),
                ("nn_bytes", C.c_uint64), ("reduce_launches", C.c_uint64), ("reduce_ms", C.c_double),
                ("reduce_bytes", C.c_uint64), ("transform_launches", C.c_uint64), ("transform_ms", C.c_double),
                ("transform_bytes", C.c_uint64), ("iterations", C.c_uint64), ("aligns", C.c_uint64),
                ("grid_launches", C.c_uint64), ("grid_ms", C.c_double), ("grid_bytes", C.c_uint64),
                ("grid_builds", C.c_uint64), ("grid_build_ms", C.c_double), ("grid_fallback_points", C.c_uint64),
                ("voxel_launches", C.c_uint64), ("voxel_ms", C.c_double), ("voxel_bytes", C.c_uint64),
                ("gicp_cov_launches", C.c_uint64), ("gicp_cov_ms", C.c_double), ("gicp_cost_launches", C.c_uint64),
                ("map_inserts", C.c_uint64), ("map_insert_ms", C.c_double), ("map_points_in", C.c_uint64),
                ("map_nn_launches", C.c_uint64), ("map_nn_ms", C.c_double),
                ("nn_timed", C.c_uint64), ("grid_timed", C.c_uint64), ("reduce_timed", C.c_uint64),
                ("grid_bounded", C.c_uint64), ("gicp_eval_ms", C.c_double), ("gicp_eval_corr", C.c_uint64), ("gicp_cov_points", C.c_uint64),
                ("targets_recognised", C.c_uint64), ("brute_bound_violations", C.c_uint64),
                ("brute_bound_worst", C.c_double), ("gicp_device_solves", C.c_uint64),
                ("grid_adopted", C.c_uint64), ("sources_adopted", C.c_uint64), ("gicp_host_solves", C.c_uint64),
                ("gicp_solver_choice", C.c_uint64), ("gicp_quadratic_solves", C.c_uint64),
                ("cov_grids_unchecked", C.c_uint64), ("cov_grids_rebuilt", C.c_uint64), ("voxel_views_direct", C.c_uint64)]


class Pose(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("quat", C.c_double * 4)]


# every symbol include/icpgpu.h declares (tests/test_abi.py checks the header against this list)
EXPORTS = [
    "icpgpu_create_abi", "icpgpu_destroy", "icpgpu_last_error", "icpgpu_version", "icpgpu_default_params_sz", "icpgpu_struct_sizes",
    "icpgpu_calibrate", "icpgpu_set_params", "icpgpu_get_params", "icpgpu_set_source", "icpgpu_set_target",
    "icpgpu_set_source_device", "icpgpu_set_target_device", "icpgpu_promote_source_to_target", "icpgpu_align", "icpgpu_align_view",
    "icpgpu_fingerprint", "icpgpu_cloud_sizes", "icpgpu_align_batch_multi_sz", "icpgpu_multi_last_error",
    "icpgpu_fitness", "icpgpu_align_batch", "icpgpu_nn", "icpgpu_reduce", "icpgpu_solve", "icpgpu_transform",
    "icpgpu_profile_reset", "icpgpu_profile_get", "icpgpu_profile_set_sampling", "icpgpu_get_stream", "icpgpu_synchronize",
    "icpgpu_voxel_grid", "icpgpu_voxel_grid_fetch", "icpgpu_voxel_grid_view", "icpgpu_set_source_voxel_filtered", "icpgpu_gicp_covariances",
    "icpgpu_gicp_quadratic_eval", "icpgpu_gicp_quadratic_sums",
    "icpgpu_pose_from_matrix", "icpgpu_pose_to_matrix", "icpgpu_pose_compose", "icpgpu_pose_inverse", "icpgpu_posegraph_create",
    "icpgpu_posegraph_destroy", "icpgpu_posegraph_set_initial_pose", "icpgpu_posegraph_push",
    "icpgpu_posegraph_num_poses", "icpgpu_posegraph_num_keyframes", "icpgpu_posegraph_get_pose",
    "icpgpu_posegraph_get_keyframe", "icpgpu_posegraph_get_edge", "icpgpu_posegraph_write_g2o",
    "icpgpu_map_set_search", "icpgpu_map_reset", "icpgpu_map_add_points", "icpgpu_map_add_source", "icpgpu_map_size", "icpgpu_map_get_points",
    "icpgpu_map_nn_target", "icpgpu_count_candidates", "icpgpu_count_candidates_read",
]

_lib = None


class IcpGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"icpgpu error {code}: {msg}")
        self.code = code


def load():
    """dlopen libicpgpu.so and declare the prototypes. Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C icpslam_amd/csrc`). icpslam_amd has no fallback path.")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.icpgpu_create_abi.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t]
    L.icpgpu_struct_sizes.argtypes = [C.POINTER(C.c_size_t)]
    L.icpgpu_struct_sizes.restype = None
    L.icpgpu_calibrate.argtypes = [vp, C.POINTER(C.c_int)]
    L.icpgpu_destroy.argtypes = [vp]
    L.icpgpu_last_error.argtypes = [vp]
    L.icpgpu_last_error.restype = C.c_char_p
    L.icpgpu_version.argtypes = []
    L.icpgpu_default_params_sz.argtypes = [C.POINTER(Params), C.c_size_t]
    L.icpgpu_default_params_sz.restype = None
    L.icpgpu_set_params.argtypes = [vp, C.POINTER(Params)]
    L.icpgpu_get_params.argtypes = [vp, C.POINTER(Params)]
    L.icpgpu_set_source.argtypes = [vp, fp, C.c_size_t]
    L.icpgpu_set_target.argtypes = [vp, fp, C.c_size_t]
    L.icpgpu_set_source_device.argtypes = [vp, vp, C.c_size_t]
    L.icpgpu_set_target_device.argtypes = [vp, vp, C.c_size_t]
    L.icpgpu_promote_source_to_target.argtypes = [vp]
    L.icpgpu_fingerprint.argtypes = [fp, C.c_size_t]
    L.icpgpu_fingerprint.restype = C.c_uint64
    L.icpgpu_cloud_sizes.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.icpgpu_align.argtypes = [vp, fp, fp, C.c_int, C.POINTER(Result)]
    L.icpgpu_align_view.argtypes = [vp, fp, C.c_int, C.POINTER(Result), C.POINTER(fp), C.POINTER(C.c_size_t)]
    L.icpgpu_fitness.argtypes = [vp, C.c_double, dp]
    L.icpgpu_align_batch.argtypes = [vp, C.c_size_t, C.POINTER(fp), C.POINTER(C.c_size_t), C.POINTER(fp),
                                     C.POINTER(C.c_size_t), C.c_int, C.POINTER(Result)]
    L.icpgpu_align_batch_multi_sz.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(Params), C.c_size_t, C.POINTER(fp),
                                              C.POINTER(C.c_size_t), C.POINTER(fp), C.POINTER(C.c_size_t), C.c_int,
                                              C.POINTER(Result), dp, C.c_int, C.c_size_t, C.c_size_t]
    L.icpgpu_multi_last_error.argtypes = []
    L.icpgpu_multi_last_error.restype = C.c_char_p
    L.icpgpu_nn.argtypes = [vp, fp, ip, fp]
    L.icpgpu_reduce.argtypes = [vp, fp, C.c_double, dp]
    L.icpgpu_solve.argtypes = [dp, dp]
    L.icpgpu_transform.argtypes = [vp, fp, fp]
    L.icpgpu_gicp_covariances.argtypes = [vp, C.c_int, dp]
    L.icpgpu_gicp_quadratic_eval.argtypes = [dp, fp, dp, dp, dp]
    L.icpgpu_gicp_quadratic_sums.argtypes = [vp, fp, dp]
    L.icpgpu_voxel_grid.argtypes = [vp, fp, C.c_size_t, C.c_float, fp, C.POINTER(C.c_size_t)]
    L.icpgpu_voxel_grid_fetch.argtypes = [vp, fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.icpgpu_voxel_grid_view.argtypes = [vp, fp, C.c_size_t, C.c_float, C.POINTER(fp), C.POINTER(C.c_size_t)]
    L.icpgpu_set_source_voxel_filtered.argtypes = [vp, fp, C.c_size_t, C.c_float, C.POINTER(C.c_size_t)]
    pp, lp = C.POINTER(Pose), C.POINTER(C.c_long)
    L.icpgpu_pose_from_matrix.argtypes = [fp, pp]
    L.icpgpu_pose_to_matrix.argtypes = [pp, fp]
    L.icpgpu_pose_compose.argtypes = [pp, pp, pp]
    L.icpgpu_pose_inverse.argtypes = [pp, pp]
    L.icpgpu_posegraph_create.argtypes = [C.POINTER(vp), C.c_double, dp]
    L.icpgpu_posegraph_destroy.argtypes = [vp]
    L.icpgpu_posegraph_set_initial_pose.argtypes = [vp, pp]
    L.icpgpu_posegraph_push.argtypes = [vp, fp, C.c_int, lp]
    L.icpgpu_posegraph_num_poses.argtypes = [vp]
    L.icpgpu_posegraph_num_keyframes.argtypes = [vp]
    L.icpgpu_posegraph_get_pose.argtypes = [vp, C.c_long, pp]
    L.icpgpu_posegraph_get_keyframe.argtypes = [vp, C.c_long, pp, lp]
    L.icpgpu_posegraph_get_edge.argtypes = [vp, C.c_long, pp]
    L.icpgpu_posegraph_write_g2o.argtypes = [vp, C.c_char_p]
    sp = C.POINTER(C.c_size_t)
    L.icpgpu_map_set_search.argtypes = [vp, C.c_int]
    L.icpgpu_map_reset.argtypes = [vp, C.c_double]
    L.icpgpu_map_add_points.argtypes = [vp, fp, C.c_size_t, fp, sp]
    L.icpgpu_map_add_source.argtypes = [vp, fp, sp]
    L.icpgpu_map_size.argtypes = [vp, sp]
    L.icpgpu_map_get_points.argtypes = [vp, fp, C.c_size_t, sp]
    L.icpgpu_map_nn_target.argtypes = [vp, fp, fp, fp, sp]
    L.icpgpu_profile_reset.argtypes = [vp]
    L.icpgpu_profile_get.argtypes = [vp, C.POINTER(Profile)]
    L.icpgpu_profile_set_sampling.argtypes = [vp, C.c_int]
    L.icpgpu_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.icpgpu_synchronize.argtypes = [vp]
    L.icpgpu_count_candidates.argtypes = [vp, C.c_int]
    L.icpgpu_count_candidates_read.argtypes = [vp, C.POINTER(C.c_uint64)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name in ("icpgpu_posegraph_num_poses", "icpgpu_posegraph_num_keyframes"):
            fn.restype = C.c_long
        elif name not in ("icpgpu_last_error", "icpgpu_default_params_sz", "icpgpu_struct_sizes", "icpgpu_fingerprint", "icpgpu_multi_last_error"):
            fn.restype = C.c_int
    # The three names icpgpu.h defines as MACROS over the sized entry points (include/icpgpu.h, "ABI rule"): the same spelling here,
    # with THESE mirrors' sizes -- a library whose structs have grown copies only what the mirrors hold.
    sizes = (C.sizeof(Params), C.sizeof(Result), C.sizeof(Profile))
    L.icpgpu_create = lambda out, device: L.icpgpu_create_abi(out, device, HEADER_VERSION, *sizes)
    L.icpgpu_default_params = lambda p: L.icpgpu_default_params_sz(p, sizes[0])
    L.icpgpu_align_batch_multi = lambda *a: L.icpgpu_align_batch_multi_sz(*a, sizes[0], sizes[1])
    _lib = L
    return L
