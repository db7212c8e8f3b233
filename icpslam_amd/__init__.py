"""icpslam_amd -- MI355X (gfx950) ICP scan-matching core behind icpslam's ICP-odometry call.

Only what the hot path needs: csrc/ (HIP kernels + C-ABI, built into libicpgpu.so), the ctypes binding,
the PCL-Registration-shaped host mirror and the synthetic scan generator.  No CPU fallback.
"""
from ._lib import GICP, GICP_INNER_EXACT, GICP_INNER_QUADRATIC, NN_AUTO, NN_BRUTE, NN_GRID, P2P_SVD, STATE_NAMES, IcpGpuError, Params, Profile, Result  # noqa: F401
from .registration import Context, GeneralizedIterativeClosestPoint, IterativeClosestPoint  # noqa: F401

__all__ = ["Context", "IterativeClosestPoint", "GeneralizedIterativeClosestPoint", "IcpGpuError", "Params", "Result", "Profile", "P2P_SVD", "GICP", "GICP_INNER_EXACT", "GICP_INNER_QUADRATIC",
           "NN_AUTO", "NN_BRUTE", "NN_GRID", "STATE_NAMES"]
