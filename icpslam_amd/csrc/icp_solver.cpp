// icp_solver.cpp -- host solve for one ICP iteration (see icp_solver.h).
#include "icp_solver.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace icpgpu {

Mat4d mat4_identity() {
  Mat4d m{};
  m[0] = m[5] = m[10] = m[15] = 1.0;
  return m;
}

Mat4d mat4_mul(const Mat4d& a, const Mat4d& b) {
  Mat4d c{};
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += a[k * 4 + row] * b[col * 4 + k];
      c[col * 4 + row] = acc;
    }
  return c;
}

void mat4_to_float(const Mat4d& a, float out[16]) {
  for (int i = 0; i < 16; ++i) out[i] = static_cast<float>(a[i]);
}

namespace {

struct Col3 {
  double v[3];
  double dot(const Col3& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  double norm() const { return std::sqrt(dot(*this)); }
};

Col3 cross(const Col3& a, const Col3& b) {
  return {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}

// Plane rotation of column pair (a, b) by (c, s): a' = c a - s b, b' = s a + c b.
inline void rotate(Col3& a, Col3& b, double c, double s) {
  for (int r = 0; r < 3; ++r) {
    const double x = a.v[r], y = b.v[r];
    a.v[r] = c * x - s * y;
    b.v[r] = s * x + c * y;
  }
}

double det_cols(const Col3 c[3]) { return c[0].dot(cross(c[1], c[2])); }

}  // namespace

// Hestenes one-sided Jacobi: orthogonalise the columns of W = A V by plane rotations accumulated in V.
void svd3x3(const double A[9], double U[9], double s[3], double V[9]) {
  Col3 w[3], v[3];
  for (int j = 0; j < 3; ++j) {
    for (int r = 0; r < 3; ++r) {
      w[j].v[r] = A[r * 3 + j];
      v[j].v[r] = (r == j) ? 1.0 : 0.0;
    }
  }
  static const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    bool rotated = false;
    for (const auto& pq : pairs) {
      Col3 &a = w[pq[0]], &b = w[pq[1]];
      const double aa = a.dot(a), bb = b.dot(b), ab = a.dot(b);
      if (ab == 0.0 || std::fabs(ab) <= 1e-17 * std::sqrt(aa * bb)) continue;
      const double zeta = (bb - aa) / (2.0 * ab);
      const double t = std::copysign(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
      const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
      rotate(a, b, c, sn);
      rotate(v[pq[0]], v[pq[1]], c, sn);
      rotated = true;
    }
    if (!rotated) break;
  }
  int order[3] = {0, 1, 2};
  double len[3] = {w[0].norm(), w[1].norm(), w[2].norm()};
  std::sort(order, order + 3, [&](int x, int y) { return len[x] > len[y]; });

  Col3 u[3], vs[3];
  for (int j = 0; j < 3; ++j) {
    const int o = order[j];
    s[j] = len[o];
    vs[j] = v[o];
    for (int r = 0; r < 3; ++r) u[j].v[r] = len[o] > 0.0 ? w[o].v[r] / len[o] : 0.0;
  }
  // rank-deficient input (planar / collinear / single-point correspondences): complete U to an orthonormal basis
  const double tiny = 1e-13 * (s[0] > 0.0 ? s[0] : 1.0);
  if (s[0] <= tiny) {
    for (int j = 0; j < 3; ++j)
      for (int r = 0; r < 3; ++r) u[j].v[r] = (r == j) ? 1.0 : 0.0;
  } else {
    if (s[1] <= tiny) {
      int k = 0;
      for (int r = 1; r < 3; ++r)
        if (std::fabs(u[0].v[r]) < std::fabs(u[0].v[k])) k = r;
      Col3 e{{0, 0, 0}};
      e.v[k] = 1.0;
      const double d = u[0].v[k];
      for (int r = 0; r < 3; ++r) e.v[r] -= d * u[0].v[r];
      const double n = e.norm();
      for (int r = 0; r < 3; ++r) u[1].v[r] = e.v[r] / n;
    }
    if (s[2] <= tiny) u[2] = cross(u[0], u[1]);
  }
  for (int j = 0; j < 3; ++j)
    for (int r = 0; r < 3; ++r) {
      U[r * 3 + j] = u[j].v[r];
      V[r * 3 + j] = vs[j].v[r];
    }
}

bool solve_umeyama(const double sums[17], Mat4d& Tk) {
  Tk = mat4_identity();
  const double n = sums[0];
  if (!(n >= 1.0)) return false;
  double mu_p[3], mu_q[3], sigma[9];
  for (int a = 0; a < 3; ++a) {
    mu_p[a] = sums[1 + a] / n;
    mu_q[a] = sums[4 + a] / n;
  }
  // Sigma = E[(q - mu_q)(p - mu_p)^T] from the raw second moment (double: 80 m ranges leave > 9 digits of headroom)
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) sigma[3 * a + b] = sums[7 + 3 * a + b] / n - mu_q[a] * mu_p[b];

  double U[9], s[3], V[9];
  svd3x3(sigma, U, s, V);
  Col3 uc[3], vc[3];
  for (int j = 0; j < 3; ++j)
    for (int r = 0; r < 3; ++r) {
      uc[j].v[r] = U[r * 3 + j];
      vc[j].v[r] = V[r * 3 + j];
    }
  const double sgn = (det_cols(uc) * det_cols(vc) < 0.0) ? -1.0 : 1.0;  // Eigen::umeyama: S(2) = -1 on a reflection
  double R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      R[r * 3 + c] = U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1] + sgn * U[r * 3 + 2] * V[c * 3 + 2];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tk[c * 4 + r] = R[r * 3 + c];
    Tk[12 + r] = mu_q[r] - (R[r * 3 + 0] * mu_p[0] + R[r * 3 + 1] * mu_p[1] + R[r * 3 + 2] * mu_p[2]);
  }
  for (double x : Tk)
    if (!std::isfinite(x)) {
      Tk = mat4_identity();
      return false;
    }
  return true;
}

ConvergenceCriteria::ConvergenceCriteria(int max_iterations, double transformation_epsilon,
                                         double euclidean_fitness_epsilon, bool force_iterations)
    : max_iterations_(max_iterations),
      rotation_threshold_(1.0 - transformation_epsilon),
      translation_threshold_(transformation_epsilon),
      mse_threshold_absolute_(1e-12),
      mse_threshold_relative_(euclidean_fitness_epsilon),
      force_(force_iterations),
      mse_prev_(DBL_MAX),
      state_(0) {}

bool ConvergenceCriteria::has_converged(int nr_iterations, const Mat4d& Tk, double mse) {
  state_ = 0;  // NOT_CONVERGED
  if (nr_iterations >= max_iterations_) {
    state_ = 1;  // ITERATIONS
    return true;
  }
  if (force_) return false;
  const double cos_angle = 0.5 * (Tk[0] + Tk[5] + Tk[10] - 1.0);
  const double translation_sqr = Tk[12] * Tk[12] + Tk[13] * Tk[13] + Tk[14] * Tk[14];
  if (cos_angle >= rotation_threshold_ && translation_sqr <= translation_threshold_) {
    state_ = 2;  // TRANSFORM
    return true;
  }
  if (std::fabs(mse - mse_prev_) < mse_threshold_absolute_) {
    state_ = 3;  // ABS_MSE
    return true;
  }
  if (std::fabs(mse - mse_prev_) / mse_prev_ < mse_threshold_relative_) {
    state_ = 4;  // REL_MSE
    return true;
  }
  mse_prev_ = mse;
  return false;
}

}  // namespace icpgpu
