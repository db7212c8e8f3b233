// icpgpu_registration.hpp -- header-only C++ shim: the PCL `Registration` protocol on top of the icpgpu C-ABI.
//
// The reference drives exactly this protocol at two call sites
//   /root/reference/src/icpslam/icp_odometer.cpp:188-201   and   src/icpslam/octree_mapper.cpp:104-117
// so switching it to the MI355X path is a one-type-name change:
//
//   - pcl::GeneralizedIterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> icp;
//   + icpgpu::IterativeClosestPoint<pcl::PointCloud<pcl::PointXYZ>> icp;
//
// CloudT is any type with a contiguous `points` container of 16-byte {x, y, z, pad} structs, `size()` and
// `resize()`: pcl::PointCloud<pcl::PointXYZ> qualifies (PointXYZ is 16 B, 16-byte aligned).
// The smart-pointer flavour (boost::shared_ptr in PCL <= 1.10, std::shared_ptr later) is a template parameter of
// the setters.  getFinalTransformation() returns Eigen::Matrix4f when Eigen is available, otherwise a POD with
// the same column-major layout.
//
// Behaviour kept from PCL: no exceptions on the data path; failure is hasConverged() == false.  The only throw
// is at construction when no gfx950 device / libicpgpu is usable (there is no CPU fallback to hide that).
#pragma once

#include <cfloat>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>

#include "icpgpu.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define ICPGPU_HAVE_EIGEN 1
#endif
#endif

namespace icpgpu {

#ifdef ICPGPU_HAVE_EIGEN
using Matrix4 = Eigen::Matrix4f;
inline Matrix4 make_matrix4(const float* colmajor) { return Eigen::Map<const Eigen::Matrix4f>(colmajor); }
#else
struct Matrix4 {  // column-major like Eigen::Matrix4f
  float m[16];
  float operator()(int r, int c) const { return m[c * 4 + r]; }
  float& operator()(int r, int c) { return m[c * 4 + r]; }
  const float* data() const { return m; }
  static Matrix4 Identity() {
    Matrix4 I;
    for (int i = 0; i < 16; ++i) I.m[i] = (i % 5 == 0) ? 1.f : 0.f;
    return I;
  }
};
inline Matrix4 make_matrix4(const float* colmajor) {
  Matrix4 M;
  std::memcpy(M.m, colmajor, sizeof(M.m));
  return M;
}
#endif

namespace detail {
// The reference constructs its registration object on the stack for every scan (icp_odometer.cpp:188); the device
// context (stream, scratch, HBM buffers) is therefore cached per thread and device, not per object.
struct CachedContext {
  icpgpu_ctx* ctx = nullptr;
  int device = -1;
  ~CachedContext() {
    if (ctx) icpgpu_destroy(ctx);
  }
};
inline icpgpu_ctx* thread_context(int device) {
  static thread_local CachedContext cache;
  if (cache.ctx && cache.device == device) return cache.ctx;
  if (cache.ctx) {
    icpgpu_destroy(cache.ctx);
    cache.ctx = nullptr;
  }
  const int rc = icpgpu_create(&cache.ctx, device);
  if (rc != ICPGPU_OK)
    throw std::runtime_error(std::string("icpgpu_create failed: ") + icpgpu_last_error(nullptr));
  cache.device = device;
  return cache.ctx;
}
}  // namespace detail

template <class CloudT>
class IterativeClosestPoint {
 public:
  explicit IterativeClosestPoint(int device = 0, icpgpu_method method = ICPGPU_P2P_SVD)
      : ctx_(detail::thread_context(device)) {
    icpgpu_default_params(&params_);
    params_.method = method;
    std::memset(&result_, 0, sizeof(result_));
    for (int i = 0; i < 16; ++i) result_.T[i] = (i % 5 == 0) ? 1.f : 0.f;
  }

  // --- the setters the reference calls (icp_odometer.cpp:189-194, octree_mapper.cpp:105-110) -------------------
  void setMaximumIterations(int n) { params_.max_iterations = n; }
  void setTransformationEpsilon(double eps) { params_.transformation_epsilon = eps; }
  void setMaxCorrespondenceDistance(double d) { params_.max_correspondence_distance = d; }
  void setEuclideanFitnessEpsilon(double eps) { params_.euclidean_fitness_epsilon = eps; }
  void setRANSACIterations(int n) { ransac_iterations_ = n; }  // the reference always passes 0 (no RANSAC rejector)
  template <class CloudPtr>
  void setInputSource(const CloudPtr& cloud) { source_ = &*cloud; }
  template <class CloudPtr>
  void setInputTarget(const CloudPtr& cloud) {
    target_ = &*cloud;
    target_from_map_ = false;
  }
  // the target is the nn cloud OctreeMap::approxNearestNeighbors just left in HBM: skips one host round trip
  void setInputTargetFromMap() { target_from_map_ = true; }

  int getMaximumIterations() const { return params_.max_iterations; }
  double getTransformationEpsilon() const { return params_.transformation_epsilon; }
  double getMaxCorrespondenceDistance() const { return params_.max_correspondence_distance; }

  // --- align(out) (icp_odometer.cpp:198, octree_mapper.cpp:114) -------------------------------------------------
  void align(CloudT& output) { align_impl(output, nullptr); }
  void align(CloudT& output, const Matrix4& guess) { align_impl(output, guess.data()); }

  Matrix4 getFinalTransformation() const { return make_matrix4(result_.T); }  // icp_odometer.cpp:199
  bool hasConverged() const { return result_.converged != 0; }                // icp_odometer.cpp:201
  double getFitnessScore(double max_range = DBL_MAX) {                        // icp_odometer.cpp:201
    double f = DBL_MAX;
    if (!aligned_ || icpgpu_fitness(ctx_, max_range, &f) != ICPGPU_OK) return DBL_MAX;
    return f;
  }
  const icpgpu_result& getResult() const { return result_; }
  const char* lastError() const { return icpgpu_last_error(ctx_); }

 private:
  using PointT = typename std::remove_reference<decltype(std::declval<CloudT>().points[0])>::type;
  static_assert(sizeof(PointT) == 16, "point type must be the 16-byte pcl::PointXYZ layout");

  void align_impl(CloudT& output, const float* guess) {
    aligned_ = false;
    result_.converged = 0;
    if (!source_ || (!target_ && !target_from_map_)) return;  // PCL: initCompute() fails, align returns, converged_ stays false
    if (icpgpu_set_params(ctx_, &params_) != ICPGPU_OK) return;
    const std::size_t ns = source_->points.size();
    if (icpgpu_set_source(ctx_, ns ? reinterpret_cast<const float*>(&source_->points[0]) : nullptr, ns) != ICPGPU_OK) return;
    if (!target_from_map_) {
      const std::size_t nt = target_->points.size();
      if (icpgpu_set_target(ctx_, nt ? reinterpret_cast<const float*>(&target_->points[0]) : nullptr, nt) != ICPGPU_OK) return;
    }
    output.points.resize(ns);
    float* out = ns ? reinterpret_cast<float*>(&output.points[0]) : nullptr;
    if (icpgpu_align(ctx_, guess, out, 0, &result_) != ICPGPU_OK) {
      result_.converged = 0;
      return;
    }
    aligned_ = true;
  }

  icpgpu_ctx* ctx_;
  icpgpu_params params_;
  icpgpu_result result_;
  const CloudT* source_ = nullptr;
  const CloudT* target_ = nullptr;
  int ransac_iterations_ = 0;
  bool aligned_ = false;
  bool target_from_map_ = false;
};

// pcl::VoxelGrid<PointT>-shaped front end for the odometer's pre-step
// (/root/reference/src/icpslam/icp_odometer.cpp:96-101):
//   pcl::VoxelGrid<pcl::PointXYZ> voxel_filter;  ->  icpgpu::VoxelGrid<pcl::PointCloud<pcl::PointXYZ>> voxel_filter;
//   voxel_filter.setInputCloud(in); voxel_filter.setLeafSize(l, l, l); voxel_filter.filter(out);
template <class CloudT>
class VoxelGrid {
 public:
  explicit VoxelGrid(int device = 0) : ctx_(detail::thread_context(device)) {}
  template <class CloudPtr>
  void setInputCloud(const CloudPtr& cloud) { input_ = &*cloud; }
  void setLeafSize(float lx, float ly, float lz) {
    if (lx != ly || ly != lz) throw std::invalid_argument("icpgpu::VoxelGrid: only cubic leaves (the reference passes one size)");
    leaf_ = lx;
  }
  void filter(CloudT& output) {
    if (!input_) return;
    const std::size_t n = input_->points.size();
    output.points.resize(n);
    std::size_t m = 0;
    const int rc = icpgpu_voxel_grid(ctx_, n ? reinterpret_cast<const float*>(&input_->points[0]) : nullptr, n, leaf_,
                                     n ? reinterpret_cast<float*>(&output.points[0]) : nullptr, &m);
    output.points.resize(rc == ICPGPU_OK ? m : 0);
  }

 private:
  icpgpu_ctx* ctx_;
  const CloudT* input_ = nullptr;
  float leaf_ = 0.1f;
};

// The mapper's map (/root/reference/src/icpslam/octree_mapper.cpp:55-90): replaces the pair
//   pcl::octree::OctreePointCloudSearch<pcl::PointXYZ>::Ptr map_octree_;  pcl::PointCloud<pcl::PointXYZ>::Ptr map_cloud_;
// Poses are the float 4x4 that pcl_ros::transformPointCloud applies (icpgpu_pose_to_matrix gives it for a Pose6DOF);
// the transform of transformCloudToPoseFrame is fused into both calls, so the mapper passes the scan in the robot frame.
template <class CloudT>
class OctreeMap {
 public:
  explicit OctreeMap(double resolution, int device = 0) : ctx_(detail::thread_context(device)), resolution_(resolution) { resetMap(); }
  void resetMap() { icpgpu_map_reset(ctx_, resolution_); }                                     // :55-59
  std::size_t addPointsToMap(const CloudT& cloud, const Matrix4& pose) {                         // :62-69 (+ :135, :152)
    std::size_t added = 0;
    const std::size_t n = cloud.points.size();
    icpgpu_map_add_points(ctx_, n ? reinterpret_cast<const float*>(&cloud.points[0]) : nullptr, n, pose.data(), &added);
    return added;
  }
  // :72-90 followed by the transform back at :146.  nearest_neighbors receives the nn cloud; it also stays in HBM as
  // the registration target (IterativeClosestPoint::setInputTargetFromMap).  Exact nearest neighbours.
  bool approxNearestNeighbors(const CloudT& cloud, const Matrix4& pose, const Matrix4& pose_inv, CloudT& nearest_neighbors) {
    const std::size_t n = cloud.points.size();
    nearest_neighbors.points.resize(n);
    std::size_t m = 0;
    if (icpgpu_set_source(ctx_, n ? reinterpret_cast<const float*>(&cloud.points[0]) : nullptr, n) != ICPGPU_OK ||
        icpgpu_map_nn_target(ctx_, pose.data(), pose_inv.data(), n ? reinterpret_cast<float*>(&nearest_neighbors.points[0]) : nullptr,
                             &m) != ICPGPU_OK)
      m = 0;
    nearest_neighbors.points.resize(m);
    return m > 0;
  }
  std::size_t size() const {
    std::size_t n = 0;
    icpgpu_map_size(ctx_, &n);
    return n;
  }
  void getMapCloud(CloudT& out) const {  // map_cloud_ for the publisher at :155
    std::size_t n = size(), m = 0;
    out.points.resize(n);
    if (icpgpu_map_get_points(ctx_, n ? reinterpret_cast<float*>(&out.points[0]) : nullptr, n, &m) != ICPGPU_OK) out.points.resize(0);
  }

 private:
  icpgpu_ctx* ctx_;
  double resolution_;
};

}  // namespace icpgpu
