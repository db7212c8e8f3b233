// icp_gicp_solver.cpp -- the host instantiation of icp_gicp_solver_impl.h (see icp_gicp_solver.h).
#include "icp_gicp_solver.h"

#include "icp_gicp_quadratic.h"

namespace icpgpu {

void gicp_apply_state(float t[16], const Vec6& x) { gicp::apply_state(t, x); }

Vec6 gicp_state_from_matrix(const float t[16]) {
  Vec6 x;
  x[0] = t[12];
  x[1] = t[13];
  x[2] = t[14];
  x[3] = std::atan2((double)t[6], (double)t[10]);
  x[4] = std::asin(-(double)t[2]);
  x[5] = std::atan2((double)t[1], (double)t[0]);
  return x;
}

void gicp_rotation_gradient(const Vec6& x, const double R[9], Vec6& g) { gicp::rotation_gradient(x, R, g); }

GicpSolve gicp_minimize(const GicpEvalFn& eval, Vec6& x, int max_inner, double gradient_tol, const GicpEval* at_x) {
  switch (gicp::minimize(eval, x, max_inner, gradient_tol, at_x)) {
    case gicp::kOk: return GicpSolve::Ok;
    case gicp::kNotEnoughPoints: return GicpSolve::NotEnoughPoints;
    case gicp::kDidNotConverge: return GicpSolve::DidNotConverge;
    default: return GicpSolve::DeviceError;
  }
}


namespace {
// quad_form_sums for the CPU this process runs on: the twelve accumulators of its main loop are vectors of four doubles
// (icp_gicp_quadratic.h) -- two SSE2 halves on the baseline target, one AVX2 register where the CPU has them; element-wise IEEE
// operations without contraction either way, so both clones return the same bits (the host's share of a scan in the quadratic mode
// is ~40 such evaluations per outer iteration).
#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
void quad_sums_for_this_cpu(const gicp::QuadForm& Q, const float* T, double* s) { gicp::quad_form_sums(Q, T, s); }

struct QuadEval {
  const gicp::QuadForm& Q;
  const float* base16;
  int evaluations = 0;
  bool operator()(const Vec6& x, GicpEval& out) {
    float T[16];
    std::memcpy(T, base16, sizeof(T));
    const gicp::Trig6 tr = gicp::trig6(x);
    gicp::apply_state(T, x, tr);
    double s[15];
    quad_sums_for_this_cpu(Q, T, s);
    gicp::eval_from_sums(tr, s, out);
    ++evaluations;
    return true;
  }
};
}  // namespace

GicpSolve gicp_minimize_quadratic(const double* sums, const float base16[16], Vec6& x, int max_inner, double gradient_tol,
                                  int* evaluations) {
  gicp::QuadForm Q;
  gicp::quad_form_load(Q, sums, base16);
  QuadEval eval{Q, base16};
  const gicp::Status st = gicp::minimize(eval, x, max_inner, gradient_tol, nullptr);
  if (evaluations) *evaluations = eval.evaluations;
  switch (st) {
    case gicp::kOk: return GicpSolve::Ok;
    case gicp::kNotEnoughPoints: return GicpSolve::NotEnoughPoints;
    case gicp::kDidNotConverge: return GicpSolve::DidNotConverge;
    default: return GicpSolve::DeviceError;
  }
}

void gicp_quadratic_eval(const double* sums, const float base16[16], const Vec6& x, GicpEval& out) {
  gicp::QuadForm Q;
  gicp::quad_form_load(Q, sums, base16);
  QuadEval eval{Q, base16};
  eval(x, out);
}

}  // namespace icpgpu
