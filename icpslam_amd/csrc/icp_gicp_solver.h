// icp_gicp_solver.h -- host side of the GICP mode (internal): the 6-parameter state <-> matrix maps and the BFGS
// minimiser PCL runs inside every outer iteration (pcl::GeneralizedIterativeClosestPoint::estimateRigidTransformationBFGS,
// reached from /root/reference/src/icpslam/icp_odometer.cpp:198 and src/icpslam/octree_mapper.cpp:114).  PCL's BFGS class
// (pcl/registration/bfgs.h) is a port of GSL's vector_bfgs2 with Fletcher's line search; restated from that published
// algorithm in icp_gicp_solver_impl.h (PCL itself is not in /root/reference), which the device solver instantiates as well.
#pragma once

#include <functional>

#include "icp_gicp_solver_impl.h"

namespace icpgpu {

using Vec6 = gicp::V6;  // tx, ty, tz, roll (about x), pitch (about y), yaw (about z)
using GicpEval = gicp::Eval;

// t <- Rz(yaw) Ry(pitch) Rx(roll) * t.R ;  t.translation += (tx, ty, tz).  float, column-major 4x4 (PCL's applyState).
void gicp_apply_state(float t[16], const Vec6& x);
// x from a float 4x4 (PCL's initial solution: atan2(T21, T22), asin(-T20), atan2(T10, T00)).
Vec6 gicp_state_from_matrix(const float t[16]);
// g[3..5] from the 3x3 sum R (row-major) through dR/d(roll, pitch, yaw) (PCL's computeRDerivative).
void gicp_rotation_gradient(const Vec6& x, const double R[9], Vec6& g);

// evaluates cost and gradient at x; returns false on a device error
using GicpEvalFn = std::function<bool(const Vec6& x, GicpEval& out)>;

enum class GicpSolve { Ok, NotEnoughPoints, DidNotConverge, DeviceError };
// runs <= max_inner BFGS steps from x (gradient tolerance 1e-2, PCL's line-search constants); x is updated in place
// at_x (nullable): value and gradient at the start point if the caller already has them (saves one evaluation)
GicpSolve gicp_minimize(const GicpEvalFn& eval, Vec6& x, int max_inner, double gradient_tol, const GicpEval* at_x = nullptr);


// The same minimisation over the QUADRATIC FORM of an outer iteration (icp_gicp_quadratic.h): no device evaluation at all, the 73
// sums the device collected once are all it needs.  sums: kGicpQuadSums double-double numbers as (hi, lo) pairs; base16: the
// column-major float 4x4 PCL calls base_transformation_.  evaluations (nullable): how many cost evaluations the run took.
GicpSolve gicp_minimize_quadratic(const double* sums, const float base16[16], Vec6& x, int max_inner, double gradient_tol,
                                  int* evaluations = nullptr);
// one evaluation of that form (tests: the algebra against a per-point evaluation)
void gicp_quadratic_eval(const double* sums, const float base16[16], const Vec6& x, GicpEval& out);

}  // namespace icpgpu
