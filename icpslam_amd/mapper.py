"""Host-side mirror of the reference's OctreeMapper for SURVEY.md section 8(f4) / BASELINE config 3: a scan is refined
against the map through the same ICP call, then grows the map.

Mirrors, call for call (/root/reference/src/icpslam/octree_mapper.cpp):
  resetMap                     :55-59    one-point-per-voxel map, octree_resolution_ = 0.5 m (:41)
  addPointsToMap               :62-69    first point of every unoccupied voxel, in input order
  approxNearestNeighbors       :72-90    nearest map point of every scan point (EXACT by default; pcl_approx_search=True gives
                                         PCL's heuristic descent, the reference's own nn cloud)
  transformCloudToPoseFrame    :92-99    pcl_ros::transformPointCloud with Pose6DOF::toTFTransform
  estimateTransformICP         :101-124  the ICP call of the hot path: 30 iterations (octree_mapper.h:56), no fitness gate
  refineTransformAndGrowMap    :133-172  the sequence below
The map, the nn cloud and the ICP all stay in HBM (icpgpu_map_*, icp_map.hip); this module only sequences the calls and
keeps the SE(3) bookkeeping (icp_posegraph.cpp, the reference's Pose6DOF rules).
"""
from __future__ import annotations

import numpy as np

from ._lib import GICP, Pose
from .registration import Context
from .sequence import pose_compose, pose_from_matrix, pose_inverse, pose_to_matrix

ICP_MAX_ITERS = 30            # octree_mapper.h:56
ICP_EPSILON = 1e-06           # octree_mapper.h:55
ICP_MAX_CORR_DIST = 1.0       # octree_mapper.h:54
OCTREE_RESOLUTION = 0.5       # octree_mapper.cpp:41


def identity_pose() -> Pose:
    return pose_from_matrix(np.eye(4, dtype=np.float32))


class OctreeMapper:
    """`OctreeMapper` of the reference, minus ROS: same method names, same order of operations."""

    def __init__(self, ctx: Context, octree_resolution: float = OCTREE_RESOLUTION, max_iterations: int = ICP_MAX_ITERS,
                 transformation_epsilon: float = ICP_EPSILON, max_correspondence_distance: float = ICP_MAX_CORR_DIST,
                 method: int | None = GICP, pcl_approx_search: bool = False):
        """method: the solver of estimateTransformICP -- GICP by default, as the reference instantiates it
        (pcl::GeneralizedIterativeClosestPoint, octree_mapper.cpp:104); P2P_SVD for point-to-point.  pcl_approx_search: collect
        the nn cloud with PCL's approxNearestSearch heuristic (what octree_mapper.cpp:84 literally calls) instead of the exact
        nearest map point."""
        self.ctx = ctx
        self.pcl_approx_search = bool(pcl_approx_search)
        self.octree_resolution = float(octree_resolution)
        self._icp = dict(max_iterations=max_iterations, transformation_epsilon=transformation_epsilon,
                         max_correspondence_distance=max_correspondence_distance)
        if method is not None:
            self._icp["method"] = method
        self.resetMap()

    # octree_mapper.cpp:55-59
    def resetMap(self):
        self.ctx.map_reset(self.octree_resolution)
        self.ctx.map_set_search(self.pcl_approx_search)

    # octree_mapper.cpp:62-69 (the cloud is given in the sensor frame together with its pose: the transform of
    # transformCloudToPoseFrame is fused into the insertion kernel)
    def addPointsToMap(self, cloud, pose: Pose | None = None) -> int:
        return self.ctx.map_add_points(cloud, None if pose is None else pose_to_matrix(pose))

    def map_cloud(self) -> np.ndarray:
        return self.ctx.map_points()

    @property
    def map_size(self) -> int:
        return self.ctx.map_size()

    # octree_mapper.cpp:72-90 + the transform back into the robot frame at :146; the result is the ICP target
    def approxNearestNeighbors(self, cloud, raw_pose: Pose, want_cloud: bool = True):
        self.ctx.set_params(self.ctx.default_params(), **self._icp)   # the target's search index is built for this gate
        self.ctx.set_source(cloud)
        return self.ctx.map_nn_target(pose_to_matrix(raw_pose), pose_to_matrix(pose_inverse(raw_pose)), want_cloud=want_cloud)

    # octree_mapper.cpp:101-124: source = the scan (already set by approxNearestNeighbors), target = the nn cloud
    def estimateTransformICP(self):
        self.ctx.set_params(self.ctx.default_params(), **self._icp)
        res = self.ctx.align()
        return bool(res["converged"]), pose_from_matrix(res["T"]), res

    # octree_mapper.cpp:133-172
    def refineTransformAndGrowMap(self, cloud, raw_pose: Pose):
        """Returns (ok, transform, refined_pose, info). ok False on the first scan (the map was empty: the scan seeds it
        at raw_pose, :137-141) and when ICP does not converge (the map is left alone, :171)."""
        if self.map_size == 0:
            added = self.addPointsToMap(cloud, raw_pose)
            return False, None, None, dict(seeded=True, added=added)
        nn = self.approxNearestNeighbors(cloud, raw_pose, want_cloud=False)
        del nn
        n_nn = self.ctx.n_target
        ok, transform, res = self.estimateTransformICP()
        if not ok:
            return False, None, None, dict(seeded=False, added=0, n_nn=n_nn, icp=res)
        refined = pose_compose(raw_pose, transform)       # `raw_pose + transform`, pose6DOF.cpp:98-105
        added = self.ctx.map_add_source(pose_to_matrix(refined))
        return True, transform, refined, dict(seeded=False, added=added, n_nn=n_nn, icp=res)
