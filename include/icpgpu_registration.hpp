// icpgpu_registration.hpp -- header-only C++ shim: the PCL `Registration` protocol on top of the icpgpu C-ABI.
//
// The reference drives exactly this protocol at two call sites
//   /root/reference/src/icpslam/icp_odometer.cpp:188-201   and   src/icpslam/octree_mapper.cpp:104-117
// so switching it to the MI355X path is a one-type-name change:
//
//   - pcl::GeneralizedIterativeClosestPoint<pcl::PointXYZ, pcl::PointXYZ> icp;
//   + icpgpu::GeneralizedIterativeClosestPoint<pcl::PointCloud<pcl::PointXYZ>> icp;   // the same solver (GICP)
//
// icpgpu::IterativeClosestPoint<Cloud> is pcl::IterativeClosestPoint's counterpart (point-to-point, SVD): the solver
// BASELINE.json's north_star specifies kernel by kernel.  Each class keeps the semantics of the PCL class it is named after.
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
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "icpgpu.h"

// -DICPGPU_SHIM_TIMING (development): wall time of the phases of align() -- 1 the uploads / recognitions, 2 icpgpu_align_view,
// 3 filling `output` -- summed in icpgpu::detail::shim_us[] (not thread-safe: for harnesses that run one callback at a time)
#if defined(ICPGPU_SHIM_TIMING)
#include <chrono>
namespace icpgpu { namespace detail {
inline double* shim_us_array() { static double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return v; }
inline void shim_mark(int k) {
  static std::chrono::steady_clock::time_point t;
  const auto now = std::chrono::steady_clock::now();
  if (k > 0) shim_us_array()[k] += std::chrono::duration<double, std::micro>(now - t).count();
  t = now;
}
} }
#define ICPGPU_SHIM_MARK(k) ::icpgpu::detail::shim_mark(k)
#else
#define ICPGPU_SHIM_MARK(k) ((void)0)
#endif

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
// The reference constructs its registration object on the stack for every scan (icp_odometer.cpp:188), and its callbacks
// run on whichever of the ROS AsyncSpinner(4) threads is free (icpslam_node.cpp:9).  The device context (stream, scratch,
// the clouds, grids and covariances in HBM) therefore lives in a per-process POOL, not in the object and not in the
// thread: an object leases a context for its lifetime and hands it back; the next object -- on any thread -- gets the most
// recently returned one, or, at align(), the idle one whose source cloud has the size of the target it is about to set
// (the odometer's previous scan: icpgpu_set_target then recognises the content and keeps cloud, grid and covariances).
// Two objects alive at once (the odometer's callback and the mapper's main loop, or two objects on one thread) hold two
// different contexts.  Contexts are never destroyed behind the caller's back: the pool is leaked at process end (no static
// destructor may run after the HIP runtime's own), release_cached_contexts() frees the idle ones on request.
struct Slot {
  icpgpu_ctx* ctx = nullptr;
  int device = 0;
  unsigned long long generation = 0;  // bumped whenever the context's clouds are replaced (only the lease holder writes it)
};
struct Pool {
  std::mutex m;
  std::vector<Slot*> idle;  // most recently returned last
};
inline Pool& pool() {
  static Pool* p = new Pool;
  return *p;
}
using ContextPtr = std::shared_ptr<Slot>;
inline void give_back(Slot* s) {
  std::lock_guard<std::mutex> g(pool().m);
  pool().idle.push_back(s);
}
// idle context of `device`: the one whose source cloud has want_source_n points if there is one, else the most recently
// returned one; nullptr if none is idle
inline Slot* take_idle(int device, std::size_t want_source_n) {
  Pool& P = pool();
  std::lock_guard<std::mutex> g(P.m);
  std::size_t pick = P.idle.size();
  for (std::size_t i = P.idle.size(); i-- > 0;) {
    if (P.idle[i]->device != device) continue;
    if (pick == P.idle.size()) pick = i;
    if (want_source_n == static_cast<std::size_t>(-1)) break;
    std::size_t ns = 0;
    icpgpu_cloud_sizes(P.idle[i]->ctx, &ns, nullptr);
    if (ns == want_source_n) {
      pick = i;
      break;
    }
  }
  if (pick == P.idle.size()) return nullptr;
  Slot* s = P.idle[pick];
  P.idle.erase(P.idle.begin() + static_cast<std::ptrdiff_t>(pick));
  return s;
}
inline ContextPtr acquire_context(int device, std::size_t want_source_n = static_cast<std::size_t>(-1)) {
  Slot* s = take_idle(device, want_source_n);
  if (!s) {
    icpgpu_ctx* raw = nullptr;
    const int rc = icpgpu_create(&raw, device);
    if (rc != ICPGPU_OK)
      throw std::runtime_error(std::string("icpgpu_create failed: ") + icpgpu_last_error(nullptr));
    s = new Slot;
    s->ctx = raw;
    s->device = device;
  }
  return ContextPtr(s, give_back);
}
// a better-matching idle context for an object about to set a target of n points; keeps `have` when there is none
inline ContextPtr rebind_for_target(const ContextPtr& have, std::size_t n_target) {
  std::size_t ns = 0;
  icpgpu_cloud_sizes(have->ctx, &ns, nullptr);
  if (ns == n_target) return have;
  Slot* s = take_idle(have->device, n_target);
  if (!s) return have;
  std::size_t ns2 = 0;
  icpgpu_cloud_sizes(s->ctx, &ns2, nullptr);
  if (ns2 != n_target) {  // just the most recently returned one: no better than what we hold
    give_back(s);
    return have;
  }
  return ContextPtr(s, give_back);
}
// the context of the OctreeMap that last built an nn cloud on this thread (setInputTargetFromMap() without an argument).
// A WEAK reference: it neither keeps the map's lease alive (the context goes back to the pool when the map dies, on whatever
// thread) nor can it hand a dead map's context to a registration object (lock() fails once the lease is over).
inline std::weak_ptr<Slot>& last_map_context() {
  static thread_local std::weak_ptr<Slot> p;
  return p;
}

// pcl::PointCloud keeps width / height / is_dense beside `points` (pcl::toROSMsg asserts width * height == size): set
// them when the cloud type has them
template <class C>
auto set_cloud_shape(C& c, std::size_t n, int) -> decltype(c.width = 0, c.height = 0, c.is_dense = true, void()) {
  c.width = static_cast<decltype(c.width)>(n);
  c.height = 1;
  c.is_dense = true;
}
template <class C>
void set_cloud_shape(C&, std::size_t, long) {}
}  // namespace detail

// destroys the pool's idle contexts (their HBM); contexts leased by live objects are untouched
inline void release_cached_contexts() {
  std::vector<detail::Slot*> idle;
  {
    std::lock_guard<std::mutex> g(detail::pool().m);
    idle.swap(detail::pool().idle);
  }
  for (detail::Slot* s : idle) {
    icpgpu_destroy(s->ctx);
    delete s;
  }
}

template <class CloudT>
class IterativeClosestPoint {
 public:
  explicit IterativeClosestPoint(int device = 0, icpgpu_method method = ICPGPU_P2P_SVD)
      : ctx_holder_(detail::acquire_context(device)), ctx_(ctx_holder_->ctx) {
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
  // the target is the nn cloud OctreeMap::approxNearestNeighbors just left in HBM (skips one host round trip): this
  // object then works on the MAP's context -- the one given, or the one of the map that last built an nn cloud on this thread
  // (no live map on this thread / a null context: there is no target -- align() then leaves hasConverged() false, like PCL's
  // initCompute() without a target)
  void setInputTargetFromMap() { setInputTargetFromMap(detail::last_map_context().lock()); }
  void setInputTargetFromMap(const detail::ContextPtr& map_context) {
    target_ = nullptr;
    target_from_map_ = static_cast<bool>(map_context);
    if (map_context) {
      ctx_holder_ = map_context;
      ctx_ = ctx_holder_->ctx;
    }
  }

  void setFitnessWithAlign(bool on) { fitness_with_align_ = on; }  // no PCL counterpart
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
    if (!aligned_) return DBL_MAX;
    // Something else may have replaced the context's clouds since (a map's context is shared with its OctreeMap, whose
    // next approxNearestNeighbors call does): put this object's clouds and transform back first.
    if (ctx_holder_->generation != generation_) {
      if (target_from_map_ || !source_ || !target_ || !upload()) return DBL_MAX;
      double sums[17];
      std::size_t n = source_->points.size();
      if (n == 0) return DBL_MAX;
      // mean squared distance of the neighbours within max_range under THIS object's transform (kernel-level entry points)
      if (fitness_sums_at(result_.T, max_range, sums) != ICPGPU_OK) return DBL_MAX;
      return sums[0] > 0.0 ? sums[16] / sums[0] : DBL_MAX;
    }
    if (fitness_with_align_ && max_range == DBL_MAX && result_.fitness == result_.fitness) return result_.fitness;
    if (icpgpu_fitness(ctx_, max_range, &f) != ICPGPU_OK) return DBL_MAX;
    return f;
  }
  const icpgpu_result& getResult() const { return result_; }
  const char* lastError() const { return icpgpu_last_error(ctx_); }

 private:
  using PointT = typename std::remove_reference<decltype(std::declval<CloudT>().points[0])>::type;
  static_assert(sizeof(PointT) == 16, "point type must be the 16-byte pcl::PointXYZ layout");

  bool upload() {
    // the TARGET first: it is usually the cloud the context still holds as the previous scan's source
    // (`*prev_cloud_ = *curr_cloud_`, icp_odometer.cpp:209), which icpgpu_set_target recognises -- no upload, the grid
    // and the GICP covariances stay -- but only as long as set_source has not replaced it
    if (!target_from_map_) {
      const std::size_t nt = target_->points.size();
      if (!bound_) {
        detail::ContextPtr better = detail::rebind_for_target(ctx_holder_, nt);
        if (better != ctx_holder_) {
          ctx_holder_ = better;
          ctx_ = ctx_holder_->ctx;
        }
        bound_ = true;
      }
      if (icpgpu_set_params(ctx_, &params_) != ICPGPU_OK) return false;
      if (icpgpu_set_target(ctx_, nt ? reinterpret_cast<const float*>(&target_->points[0]) : nullptr, nt) != ICPGPU_OK) return false;
    } else if (icpgpu_set_params(ctx_, &params_) != ICPGPU_OK) {
      return false;
    }
    const std::size_t ns = source_->points.size();
    return icpgpu_set_source(ctx_, ns ? reinterpret_cast<const float*>(&source_->points[0]) : nullptr, ns) == ICPGPU_OK;
  }

  // fitness through the kernel-level entry points, for a transform that is not the context's last one
  int fitness_sums_at(const float* T, double max_range, double sums[17]) {
    const std::size_t n = source_->points.size();
    std::unique_ptr<int32_t[]> idx(new int32_t[n]);
    std::unique_ptr<float[]> d2(new float[n]);
    int rc = icpgpu_nn(ctx_, T, idx.get(), d2.get());
    if (rc != ICPGPU_OK) return rc;
    // PCL compares SQUARED distances with max_range; icpgpu_reduce takes the distance
    return icpgpu_reduce(ctx_, T, max_range >= 1e36 ? 1e18 : std::sqrt(max_range), sums);
  }

  void align_impl(CloudT& output, const float* guess) {
    aligned_ = false;
    result_.converged = 0;
    if (!source_ || (!target_ && !target_from_map_)) return;  // PCL: initCompute() fails, align returns, converged_ stays false
    ICPGPU_SHIM_MARK(-1);
    if (!upload()) return;
    const std::size_t ns = source_->points.size();
    generation_ = ++ctx_holder_->generation;
    // getFitnessScore() nearly always follows (icp_odometer.cpp:201): evaluated inside align it is one more sweep queued behind
    // the last iteration instead of a call of its own (setFitnessWithAlign(false) for callers that never ask: octree_mapper.cpp:117).
    // The aligned cloud comes as a view of the context's pinned staging buffer (the transform kernel writes it there while the
    // fitness sweep is still to run): `assign` fills `output` in one pass -- resize() + a copy would touch it twice.
    const float* view = nullptr;
    std::size_t nv = 0;
    ICPGPU_SHIM_MARK(1);
    if (icpgpu_align_view(ctx_, guess, fitness_with_align_ ? 1 : 0, &result_, &view, &nv) != ICPGPU_OK || nv != ns) {
      result_.converged = 0;
      output.points.resize(ns);  // (PCL sizes the output before it computes anything)
      detail::set_cloud_shape(output, ns, 0);
      return;
    }
    ICPGPU_SHIM_MARK(2);
    const PointT* first = reinterpret_cast<const PointT*>(view);
    output.points.assign(first, first + nv);
    detail::set_cloud_shape(output, ns, 0);
    aligned_ = true;
    ICPGPU_SHIM_MARK(3);
  }

  detail::ContextPtr ctx_holder_;
  icpgpu_ctx* ctx_;
  unsigned long long generation_ = 0;

 protected:
  icpgpu_params params_;  // (GeneralizedIterativeClosestPoint sets its solver options here)

 private:
  icpgpu_result result_;
  const CloudT* source_ = nullptr;
  const CloudT* target_ = nullptr;
  int ransac_iterations_ = 0;
  bool aligned_ = false;
  bool target_from_map_ = false;
  bool fitness_with_align_ = true;
  bool bound_ = false;  // the pool has been asked once for the context that suits this object's target
};

// pcl::GeneralizedIterativeClosestPoint<PointXYZ, PointXYZ>'s counterpart -- the class the reference instantiates at
// icp_odometer.cpp:188 and octree_mapper.cpp:104: same protocol, method = ICPGPU_GICP (plane-to-plane cost, BFGS inner solver,
// PCL's constructor defaults: 20 neighbours, gicp_epsilon 1e-3, rotation_epsilon 2e-3, 20 inner iterations).
template <class CloudT>
class GeneralizedIterativeClosestPoint : public IterativeClosestPoint<CloudT> {
 public:
  explicit GeneralizedIterativeClosestPoint(int device = 0) : IterativeClosestPoint<CloudT>(device, ICPGPU_GICP) {}
  // NOT a PCL method: the inner minimisation on the quadratic form of an outer iteration (icpgpu.h: icpgpu_gicp_inner) -- 2-3x the
  // scans/s of the reference's pipeline, results within the stated tolerance of the default's instead of on its bits.  An
  // unchanged call site opts in with ICPGPU_GICP_INNER=quadratic in the environment.
  void setQuadraticInnerSolver(bool on) { this->params_.gicp_inner = on ? ICPGPU_GICP_INNER_QUADRATIC : ICPGPU_GICP_INNER_EXACT; }
};

// pcl::VoxelGrid<PointT>-shaped front end for the odometer's pre-step
// (/root/reference/src/icpslam/icp_odometer.cpp:96-101):
//   pcl::VoxelGrid<pcl::PointXYZ> voxel_filter;  ->  icpgpu::VoxelGrid<pcl::PointCloud<pcl::PointXYZ>> voxel_filter;
//   voxel_filter.setInputCloud(in); voxel_filter.setLeafSize(l, l, l); voxel_filter.filter(out);
template <class CloudT>
class VoxelGrid {
 public:
  explicit VoxelGrid(int device = 0) : ctx_holder_(detail::acquire_context(device)), ctx_(ctx_holder_->ctx) {}
  template <class CloudPtr>
  void setInputCloud(const CloudPtr& cloud) { input_ = &*cloud; }
  void setLeafSize(float lx, float ly, float lz) {
    if (lx != ly || ly != lz) throw std::invalid_argument("icpgpu::VoxelGrid: only cubic leaves (the reference passes one size)");
    leaf_ = lx;
  }
  void filter(CloudT& output) {
    if (!input_) return;
    const std::size_t n = input_->points.size();
    std::size_t m = 0;
    // the result as a view of the context's pinned staging buffer (the points arrive there in front of the voxel count the call
    // waits for): `assign` sizes and fills `output` in one pass -- sizing it for the worst case first would value-initialise n points
    // (3.2 MB for a raw 200k-point scan) to receive a tenth of them, and a fetch of its own is a second round trip to the device
    typedef typename std::remove_reference<decltype(output.points[0])>::type PointT;
    static_assert(sizeof(PointT) == 16, "icpgpu: 16-byte points (pcl::PointXYZ)");
    const float* view = nullptr;
    const int rc = icpgpu_voxel_grid_view(ctx_, n ? reinterpret_cast<const float*>(&input_->points[0]) : nullptr, n, leaf_, &view, &m);
    if (rc == ICPGPU_OK && m) {
      const PointT* first = reinterpret_cast<const PointT*>(view);
      output.points.assign(first, first + m);
    } else {
      output.points.resize(0);
    }
    detail::set_cloud_shape(output, output.points.size(), 0);
  }

 private:
  detail::ContextPtr ctx_holder_;
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
  explicit OctreeMap(double resolution, int device = 0)
      : ctx_holder_(detail::acquire_context(device)), ctx_(ctx_holder_->ctx), resolution_(resolution) { resetMap(); }
  const detail::ContextPtr& context() const { return ctx_holder_; }  // for IterativeClosestPoint::setInputTargetFromMap(map.context())
  // :55-59.  The context comes from the pool and may have served another map before: the search mode is THIS object's
  // (exact unless setPclApproximateSearch(true) was called on it), re-applied with every reset.
  void resetMap() {
    icpgpu_map_reset(ctx_, resolution_);
    icpgpu_map_set_search(ctx_, pcl_approx_ ? ICPGPU_MAP_SEARCH_PCL_APPROX : ICPGPU_MAP_SEARCH_EXACT);
  }
  // which neighbour approxNearestNeighbors collects: false (default) = the EXACT nearest map point; true = a restatement of
  // PCL's approxNearestSearch heuristic (icpgpu.h: icpgpu_map_set_search; unpinned against a PCL build like the rest)
  void setPclApproximateSearch(bool on) {
    pcl_approx_ = on;
    icpgpu_map_set_search(ctx_, on ? ICPGPU_MAP_SEARCH_PCL_APPROX : ICPGPU_MAP_SEARCH_EXACT);
  }
  std::size_t addPointsToMap(const CloudT& cloud, const Matrix4& pose) {                         // :62-69 (+ :135, :152)
    std::size_t added = 0;
    const std::size_t n = cloud.points.size();
    icpgpu_map_add_points(ctx_, n ? reinterpret_cast<const float*>(&cloud.points[0]) : nullptr, n, pose.data(), &added);
    return added;
  }
  // :72-90 followed by the transform back at :146.  nearest_neighbors receives the nn cloud; it also stays in HBM as
  // the registration target (IterativeClosestPoint::setInputTargetFromMap).  Exact nearest neighbours unless
  // setPclApproximateSearch(true).
  bool approxNearestNeighbors(const CloudT& cloud, const Matrix4& pose, const Matrix4& pose_inv, CloudT& nearest_neighbors) {
    const std::size_t n = cloud.points.size();
    nearest_neighbors.points.resize(n);
    std::size_t m = 0;
    if (icpgpu_set_source(ctx_, n ? reinterpret_cast<const float*>(&cloud.points[0]) : nullptr, n) != ICPGPU_OK ||
        icpgpu_map_nn_target(ctx_, pose.data(), pose_inv.data(), n ? reinterpret_cast<float*>(&nearest_neighbors.points[0]) : nullptr,
                             &m) != ICPGPU_OK)
      m = 0;
    nearest_neighbors.points.resize(m);
    detail::set_cloud_shape(nearest_neighbors, m, 0);
    ++ctx_holder_->generation;  // the context's source / target are the map's now
    detail::last_map_context() = ctx_holder_;
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
    detail::set_cloud_shape(out, out.points.size(), 0);
  }

 private:
  detail::ContextPtr ctx_holder_;
  icpgpu_ctx* ctx_;
  double resolution_;
  bool pcl_approx_ = false;
};

}  // namespace icpgpu
