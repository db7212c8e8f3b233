// icp_gicp_solver.cpp -- see icp_gicp_solver.h.
#include "icp_gicp_solver.h"

#include <cfloat>
#include <cmath>
#include <limits>

namespace icpgpu {

void gicp_apply_state(float t[16], const Vec6& x) {
  // Eigen: Matrix3f R = AngleAxisf(yaw, Z) * AngleAxisf(pitch, Y) * AngleAxisf(roll, X) -- a quaternion product
  struct Qf {
    float w, x, y, z;
  };
  auto mul = [](const Qf& a, const Qf& b) {
    return Qf{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  };
  const float hr = 0.5f * (float)x[3], hp = 0.5f * (float)x[4], hy = 0.5f * (float)x[5];
  const Qf qx{std::cos(hr), std::sin(hr), 0.f, 0.f}, qy{std::cos(hp), 0.f, std::sin(hp), 0.f}, qz{std::cos(hy), 0.f, 0.f, std::sin(hy)};
  const Qf q = mul(mul(qz, qy), qx);
  const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  const float R[3][3] = {{1.f - (tyy + tzz), txy - twz, txz + twy},
                         {txy + twz, 1.f - (txx + tzz), tyz - twx},
                         {txz - twy, tyz + twx, 1.f - (txx + tyy)}};
  float out[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      for (int k = 0; k < 3; ++k) acc += R[r][k] * t[c * 4 + k];
      out[r][c] = acc;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) t[c * 4 + r] = out[r][c];
  t[12] += (float)x[0];
  t[13] += (float)x[1];
  t[14] += (float)x[2];
}

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

void gicp_rotation_gradient(const Vec6& x, const double R[9], Vec6& g) {
  const double cr = std::cos(x[3]), sr = std::sin(x[3]), cp = std::cos(x[4]), sp = std::sin(x[4]);
  const double cy = std::cos(x[5]), sy = std::sin(x[5]);
  // derivatives of Rz(yaw) Ry(pitch) Rx(roll), row-major
  const double d_roll[9] = {0, sr * sy + cr * cy * sp, cr * sy - cy * sr * sp,
                            0, -cy * sr + cr * sy * sp, -cr * cy - sr * sy * sp,
                            0, cr * cp, -cp * sr};
  const double d_pitch[9] = {-cy * sp, cy * cp * sr, cr * cy * cp,
                             -sy * sp, cp * sr * sy, cr * cp * sy,
                             -cp, -sr * sp, -cr * sp};
  const double d_yaw[9] = {-cp * sy, -cr * cy - sr * sy * sp, cy * sr - cr * sy * sp,
                           cy * cp, -cr * sy + cy * sr * sp, sr * sy + cr * cy * sp,
                           0, 0, 0};
  auto inner = [&](const double* D) {  // PCL's matricesInnerProd: sum_ij D(j,i) * R(i,j)
    double s = 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) s += D[3 * j + i] * R[3 * i + j];
    return s;
  };
  g[3] = inner(d_roll);
  g[4] = inner(d_pitch);
  g[5] = inner(d_yaw);
}

namespace {

double dot(const Vec6& a, const Vec6& b) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += a[i] * b[i];
  return s;
}
double norm(const Vec6& a) { return std::sqrt(dot(a, a)); }

// ---- Fletcher line search pieces (GSL linear_minimize.c) ----
int solve_quadratic(double a, double b, double c, double& r0, double& r1) {
  if (a == 0) {
    if (b == 0) return 0;
    r0 = -c / b;
    return 1;
  }
  const double disc = b * b - 4 * a * c;
  if (disc > 0) {
    if (b == 0) {
      const double r = std::sqrt(-c / a);
      r0 = -r;
      r1 = r;
    } else {
      const double tmp = -0.5 * (b + (b > 0 ? 1.0 : -1.0) * std::sqrt(disc));
      const double a1 = tmp / a, a2 = c / tmp;
      r0 = std::min(a1, a2);
      r1 = std::max(a1, a2);
    }
    return 2;
  }
  if (disc == 0) {
    r0 = r1 = -0.5 * b / a;
    return 2;
  }
  return 0;
}

double quad_min(double f0, double fp0, double f1, double zl, double zh) {
  auto val = [&](double z) { return f0 + z * (fp0 + z * (f1 - f0 - fp0)); };
  double zmin = zl, fmin = val(zl);
  if (val(zh) < fmin) {
    zmin = zh;
    fmin = val(zh);
  }
  const double curv = 2 * (f1 - f0 - fp0);
  if (curv > 0) {
    const double z = -fp0 / curv;
    if (z > zl && z < zh && val(z) < fmin) zmin = z;
  }
  return zmin;
}

double cubic_min(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double c2 = 3 * (f1 - f0) - 2 * fp0 - fp1, c3 = fp0 + fp1 - 2 * (f1 - f0);
  auto val = [&](double z) { return f0 + z * (fp0 + z * (c2 + z * c3)); };
  double zmin = zl, fmin = val(zl);
  auto check = [&](double z) {
    const double y = val(z);
    if (y < fmin) {
      zmin = z;
      fmin = y;
    }
  };
  check(zh);
  double z0 = 0, z1 = 0;
  const int n = solve_quadratic(3 * c3, 2 * c2, fp0, z0, z1);
  if (n >= 1 && z0 > zl && z0 < zh) check(z0);
  if (n == 2 && z1 > zl && z1 < zh) check(z1);
  return zmin;
}

double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) {
  double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
  if (ymin > ymax) std::swap(ymin, ymax);
  const double y = std::isnan(fpb) ? quad_min(fa, fpa * (b - a), fb, ymin, ymax)
                                   : cubic_min(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax);  // order 3
  return a + y * (b - a);
}

// the objective along the current search line, with GSL's value caches
class LineFunction {
 public:
  LineFunction(const GicpEvalFn& eval) : eval_(eval) {}
  void reset(const Vec6& x, double f, const Vec6& g, const Vec6& p) {
    x0_ = x;
    p_ = p;
    x_alpha_ = x;
    g_alpha_ = g;
    f_alpha_ = f;
    df_alpha_ = dot(g, p);
    f_key_ = df_key_ = x_key_ = g_key_ = 0.0;
  }
  bool ok() const { return ok_; }
  double f(double alpha) {
    if (alpha == f_key_) return f_alpha_;
    move(alpha);
    // One device reduction yields the value AND the gradient (same sums), so the gradient is cached here as well: GSL
    // asks for df(alpha) right after f(alpha) on every accepted trial point, which would otherwise be a second launch
    // producing bit for bit the same numbers.
    GicpEval e;
    ok_ = ok_ && eval_(x_alpha_, true, e);
    f_alpha_ = e.f;
    f_key_ = alpha;
    g_alpha_ = e.g;
    g_key_ = alpha;
    return f_alpha_;
  }
  double df(double alpha) {
    if (alpha == df_key_) return df_alpha_;
    move(alpha);
    if (alpha != g_key_) {
      GicpEval e;
      ok_ = ok_ && eval_(x_alpha_, true, e);
      g_alpha_ = e.g;
      g_key_ = alpha;
    }
    df_alpha_ = dot(g_alpha_, p_);
    df_key_ = alpha;
    return df_alpha_;
  }
  void fdf(double alpha, double& f_out, double& df_out) {
    if (alpha == f_key_ || alpha == df_key_) {
      f_out = f(alpha);
      df_out = df(alpha);
      return;
    }
    move(alpha);
    GicpEval e;
    ok_ = ok_ && eval_(x_alpha_, true, e);
    f_alpha_ = e.f;
    g_alpha_ = e.g;
    f_key_ = g_key_ = alpha;
    df_alpha_ = dot(g_alpha_, p_);
    df_key_ = alpha;
    f_out = f_alpha_;
    df_out = df_alpha_;
  }
  const Vec6& x_alpha() const { return x_alpha_; }
  const Vec6& g_alpha() const { return g_alpha_; }

 private:
  void move(double alpha) {
    if (alpha == x_key_) return;
    for (int i = 0; i < 6; ++i) x_alpha_[i] = x0_[i] + alpha * p_[i];
    x_key_ = alpha;
  }
  const GicpEvalFn& eval_;
  Vec6 x0_{}, p_{}, x_alpha_{}, g_alpha_{};
  double f_alpha_ = 0, df_alpha_ = 0, f_key_ = 0, df_key_ = 0, x_key_ = 0, g_key_ = 0;
  bool ok_ = true;
};

enum class Line { Found, NoProgress };

Line line_search(LineFunction& fn, double alpha1, double& alpha_out) {
  constexpr double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5;
  constexpr int bracket_iters = 100, section_iters = 100;
  const double kNaN = std::numeric_limits<double>::quiet_NaN();
  double f0, fp0;
  fn.fdf(0.0, f0, fp0);
  double alpha = alpha1, alpha_prev = 0.0, falpha, fpalpha, falpha_prev = f0, fpalpha_prev = fp0;
  double a = 0.0, b = alpha, fa = f0, fb = 0.0, fpa = fp0, fpb = 0.0;
  int i = 0;
  while (i++ < bracket_iters) {
    falpha = fn.f(alpha);
    if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
      a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
      b = alpha; fb = falpha; fpb = kNaN;
      break;
    }
    fpalpha = fn.df(alpha);
    if (std::fabs(fpalpha) <= -sigma * fp0) {
      alpha_out = alpha;
      return Line::Found;
    }
    if (fpalpha >= 0) {
      a = alpha; fa = falpha; fpa = fpalpha;
      b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
      break;
    }
    const double delta = alpha - alpha_prev;
    const double next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta);
    alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
    alpha = next;
  }
  while (i++ < section_iters) {
    const double delta = b - a;
    alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta);
    falpha = fn.f(alpha);
    if ((a - alpha) * fpa <= DBL_EPSILON) return Line::NoProgress;
    if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
      b = alpha; fb = falpha; fpb = kNaN;
    } else {
      fpalpha = fn.df(alpha);
      if (std::fabs(fpalpha) <= -sigma * fp0) {
        alpha_out = alpha;
        return Line::Found;
      }
      if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
        b = a; fb = fa; fpb = fpa;
      }
      a = alpha; fa = falpha; fpa = fpalpha;
    }
  }
  return Line::Found;  // iteration budget exhausted: alpha_out keeps the caller's 0.0, as in GSL
}

}  // namespace

GicpSolve gicp_minimize(const GicpEvalFn& eval, Vec6& x, int max_inner, double gradient_tol, const GicpEval* at_x) {
  GicpEval e0;
  if (at_x) e0 = *at_x;  // the caller has just evaluated value and gradient at x
  else if (!eval(x, true, e0)) return GicpSolve::DeviceError;
  double f = e0.f;
  Vec6 g = e0.g, x0 = x, g0 = g, p;
  double g0norm = norm(g0);
  for (int i = 0; i < 6; ++i) p[i] = -g[i] / g0norm;
  double pnorm = norm(p), fp0 = -g0norm, delta_f = 0.0;
  LineFunction line(eval);
  line.reset(x0, f, g0, p);

  int inner = 0;
  bool no_progress = false, success = false;
  do {
    ++inner;
    // ---- one vector_bfgs2 iteration ----
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0 || std::isnan(g0norm)) {
      no_progress = true;
      break;
    }
    const double f_before = f;
    double alpha1 = 1.0;  // |step_size|
    if (delta_f < 0) alpha1 = std::min(1.0, 2.0 * std::max(-delta_f, 10 * DBL_EPSILON * std::fabs(f_before)) / (-fp0));
    double alpha = 0.0;
    const Line ls = line_search(line, alpha1, alpha);
    if (!line.ok()) return GicpSolve::DeviceError;
    if (ls == Line::NoProgress) {
      no_progress = true;
      break;
    }
    double df_unused;
    line.fdf(alpha, f, df_unused);
    if (!line.ok()) return GicpSolve::DeviceError;
    x = line.x_alpha();
    g = line.g_alpha();
    delta_f = f - f_before;
    // memoryless BFGS direction p' = g - A dx - B dg
    Vec6 dx, dg;
    for (int i = 0; i < 6; ++i) {
      dx[i] = x[i] - x0[i];
      dg[i] = g[i] - g0[i];
    }
    const double dxg = dot(dx, g), dgg = dot(dg, g), dxdg = dot(dx, dg), dgn = norm(dg);
    double A = 0, B = 0;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgn * dgn / dxdg) * B + dgg / dxdg;
    }
    for (int i = 0; i < 6; ++i) p[i] = g[i] - A * dx[i] - B * dg[i];
    g0 = g;
    x0 = x;
    g0norm = norm(g0);
    pnorm = norm(p);
    const double dir = dot(p, g) >= 0.0 ? -1.0 : 1.0;
    for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
    pnorm = norm(p);
    fp0 = dot(p, g0);
    line.reset(x0, f, g0, p);
    // ---- PCL: testGradient ----
    success = norm(g) < gradient_tol;
  } while (!success && inner < max_inner);
  if (no_progress || success || inner == max_inner) return GicpSolve::Ok;
  return GicpSolve::DidNotConverge;
}

}  // namespace icpgpu
