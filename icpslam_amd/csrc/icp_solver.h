// icp_solver.h -- host-side pieces of the ICP iteration (internal): the 3x3 SVD / rigid-transform solve (a5),
// the convergence test (a7) and 4x4 composition.  north_star keeps these on the host; they run in double.
//
// Restates, for the call sites /root/reference/src/icpslam/icp_odometer.cpp:189-190,198-201 and
// src/icpslam/octree_mapper.cpp:105-106,114-117:
//   pcl::registration::TransformationEstimationSVD -> Eigen::umeyama(src, dst, /*scaling*/false)
//   pcl::registration::DefaultConvergenceCriteria::hasConverged()
#pragma once

#include <array>
#include <cstdint>

namespace icpgpu {

using Mat4d = std::array<double, 16>;  // column-major, like Eigen::Matrix4d

Mat4d mat4_identity();
Mat4d mat4_mul(const Mat4d& a, const Mat4d& b);
void mat4_to_float(const Mat4d& a, float out[16]);

// sums = {n, sum p(3), sum q(3), sum q p^T (9, row-major, row = q), sum d2}. Returns false when n < 1 or the
// result is not finite (Tk is then identity).
bool solve_umeyama(const double sums[17], Mat4d& Tk);

// A = U diag(s) V^T for a row-major 3x3, singular values descending, U and V orthonormal.
void svd3x3(const double A[9], double U[9], double s[3], double V[9]);

// pcl::registration::DefaultConvergenceCriteria with the settings IterativeClosestPoint applies:
//   max iterations, rotation threshold 1 - transformation_epsilon (on cos theta), translation threshold
//   transformation_epsilon (on |t|^2), absolute MSE 1e-12, relative MSE euclidean_fitness_epsilon,
//   max_iterations_similar_transforms = 0.
class ConvergenceCriteria {
 public:
  ConvergenceCriteria(int max_iterations, double transformation_epsilon, double euclidean_fitness_epsilon,
                      bool force_iterations);
  // nr_iterations = iterations completed including this one; Tk = this iteration's incremental transform;
  // mse = mean squared distance of this iteration's correspondences. Returns true when iteration must stop.
  bool has_converged(int nr_iterations, const Mat4d& Tk, double mse);
  int state() const { return state_; }

 private:
  int max_iterations_;
  double rotation_threshold_, translation_threshold_, mse_threshold_absolute_, mse_threshold_relative_;
  bool force_;
  double mse_prev_;
  int state_;
};

}  // namespace icpgpu
