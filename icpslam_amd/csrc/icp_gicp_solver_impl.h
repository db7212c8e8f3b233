// icp_gicp_solver_impl.h -- GICP's inner solver as ONE piece of source for the host and the device (INTERNAL).
//
// What PCL runs inside every outer iteration of pcl::GeneralizedIterativeClosestPoint (reached from
// /root/reference/src/icpslam/icp_odometer.cpp:198 and src/icpslam/octree_mapper.cpp:114):
// estimateRigidTransformationBFGS -- a port of GSL's vector_bfgs2 with Fletcher's line search (pcl/registration/bfgs.h) over the
// six parameters (tx, ty, tz, roll, pitch, yaw), with PCL's applyState / computeRDerivative as the state maps.  Restated from the
// published algorithm (PCL is not in /root/reference).  The test oracle holds an independent restatement in plain C.
//
// Since round 4 the solver runs in two places: on the host (icp_gicp_solver.cpp: the fallback path, one device evaluation per
// call) and INSIDE the resident kernel gicp_solve_kernel (icp_gicp.hip: the whole BFGS run of an outer iteration without a
// host round trip).  Both instantiate the templates below -- same operations in the same order, IEEE float64 (and float32 in
// apply_state), no contraction (-ffp-contract=off), sines and cosines from icp_trig.h -- so both produce the same bits.
// The evaluator is a template parameter: bool eval(const V6& x, Eval& out) fills value and gradient (one device reduction
// yields both), false = device error.
#pragma once

#include <cfloat>
#include <cmath>

#include "icp_trig.h"

namespace icpgpu {
namespace gicp {

struct V6 {  // tx, ty, tz, roll (about x), pitch (about y), yaw (about z)
  double v[6];
  ICPGPU_HDI double& operator[](int i) { return v[i]; }
  ICPGPU_HDI const double& operator[](int i) const { return v[i]; }
  ICPGPU_HDI void fill(double a) {
    _Pragma("unroll")
    for (int i = 0; i < 6; ++i) v[i] = a;
  }
};
struct Eval {
  double f;
  V6 g;
};
enum Status { kOk = 0, kNotEnoughPoints = 1, kDidNotConverge = 2, kDeviceError = 3 };

ICPGPU_HDI inline double dmin(double a, double b) { return b < a ? b : a; }  // std::min / std::max, spelled out
ICPGPU_HDI inline double dmax(double a, double b) { return a < b ? b : a; }

// The six sine / cosine pairs of a state: of the float half angles (applyState's quaternions) and of the double angles
// (computeRDerivative).  They depend on x only, so an evaluation computes them once -- the device six lanes at a time
// (icp_gicp.hip), the host one after the other; the same correctly rounded values either way (icp_trig.h).
struct Trig6 {
  float shr, chr, shp, chp, shy, chy;  // sin / cos of 0.5f * (float) roll, pitch, yaw
  double sr, cr, sp, cp, sy, cy;       // sin / cos of roll, pitch, yaw
};
ICPGPU_HDI inline void trig6_arguments(const V6& x, double a[6]) {  // the six arguments, as doubles (a float converts exactly)
  a[0] = (double)(0.5f * (float)x[3]);
  a[1] = (double)(0.5f * (float)x[4]);
  a[2] = (double)(0.5f * (float)x[5]);
  a[3] = x[3];
  a[4] = x[4];
  a[5] = x[5];
}
ICPGPU_HDI inline void trig6_from(const double s[6], const double c[6], Trig6& t) {  // sincos_cr results of the six arguments
  t.shr = (float)s[0]; t.chr = (float)c[0];
  t.shp = (float)s[1]; t.chp = (float)c[1];
  t.shy = (float)s[2]; t.chy = (float)c[2];
  t.sr = s[3]; t.cr = c[3];
  t.sp = s[4]; t.cp = c[4];
  t.sy = s[5]; t.cy = c[5];
}
ICPGPU_HDI inline Trig6 trig6(const V6& x) {
  double a[6], s[6], c[6];
  trig6_arguments(x, a);
  _Pragma("unroll")
  for (int i = 0; i < 6; ++i) trig::sincos_cr(a[i], &s[i], &c[i]);
  Trig6 t;
  trig6_from(s, c, t);
  return t;
}

// t <- Rz(yaw) Ry(pitch) Rx(roll) * t.R ;  t.translation += (tx, ty, tz).  float, column-major 4x4 (PCL's applyState:
// Eigen's AngleAxisf(yaw, Z) * AngleAxisf(pitch, Y) * AngleAxisf(roll, X) is a quaternion product)
ICPGPU_HDI inline void apply_state(float t[16], const V6& x, const Trig6& tr) {
  struct Qf {
    float w, x, y, z;
  };
  auto mul = [](const Qf& a, const Qf& b) {
    return Qf{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  };
  const Qf qx{tr.chr, tr.shr, 0.f, 0.f}, qy{tr.chp, 0.f, tr.shp, 0.f}, qz{tr.chy, 0.f, 0.f, tr.shy};
  const Qf q = mul(mul(qz, qy), qx);
  const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  const float R[3][3] = {{1.f - (tyy + tzz), txy - twz, txz + twy},
                         {txy + twz, 1.f - (txx + tzz), tyz - twx},
                         {txz - twy, tyz + twx, 1.f - (txx + tyy)}};
  float out[3][3];
  _Pragma("unroll")
  for (int r = 0; r < 3; ++r)
    _Pragma("unroll")
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      _Pragma("unroll")
      for (int k = 0; k < 3; ++k) acc += R[r][k] * t[c * 4 + k];
      out[r][c] = acc;
    }
  _Pragma("unroll")
  for (int r = 0; r < 3; ++r)
    _Pragma("unroll")
    for (int c = 0; c < 3; ++c) t[c * 4 + r] = out[r][c];
  t[12] += (float)x[0];
  t[13] += (float)x[1];
  t[14] += (float)x[2];
}
ICPGPU_HDI inline void apply_state(float t[16], const V6& x) { apply_state(t, x, trig6(x)); }

// g[3..5] from the 3x3 sum R (row-major) through dR/d(roll, pitch, yaw) (PCL's computeRDerivative)
ICPGPU_HDI inline void rotation_gradient(const Trig6& tr, const double R[9], V6& g) {
  const double cr = tr.cr, sr = tr.sr, cp = tr.cp, sp = tr.sp, cy = tr.cy, sy = tr.sy;
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
    _Pragma("unroll")
    for (int i = 0; i < 3; ++i)
      _Pragma("unroll")
      for (int j = 0; j < 3; ++j) s += D[3 * j + i] * R[3 * i + j];
    return s;
  };
  g[3] = inner(d_roll);
  g[4] = inner(d_pitch);
  g[5] = inner(d_yaw);
}
ICPGPU_HDI inline void rotation_gradient(const V6& x, const double R[9], V6& g) { rotation_gradient(trig6(x), R, g); }

// value and gradient from the 14 sums of an evaluation: s[0] = m, s[1] = sum r^T M r, s[2..4] = sum M r,
// s[5..13] = sum (base p)(M r)^T (row-major).  m < 1: value 0, gradient 0 (the caller deals with "not enough points").
ICPGPU_HDI inline void eval_from_sums(const Trig6& tr, const double* s, Eval& out) {
  if (!(s[0] >= 1.0)) {
    out.f = 0.0;
    out.g.fill(0.0);
    return;
  }
  out.f = s[1] / s[0];
  const double sc = 2.0 / s[0];
  double Rm[9];
  _Pragma("unroll")
  for (int k = 0; k < 3; ++k) out.g[k] = s[2 + k] * sc;
  _Pragma("unroll")
  for (int k = 0; k < 9; ++k) Rm[k] = s[5 + k] * sc;
  rotation_gradient(tr, Rm, out.g);
}
ICPGPU_HDI inline void eval_from_sums(const V6& x, const double* s, Eval& out) { eval_from_sums(trig6(x), s, out); }

ICPGPU_HDI inline double dot(const V6& a, const V6& b) {
  double s = 0;
  _Pragma("unroll")
  for (int i = 0; i < 6; ++i) s += a[i] * b[i];
  return s;
}
ICPGPU_HDI inline double norm(const V6& a) { return sqrt(dot(a, a)); }

// ---- Fletcher line search pieces (GSL linear_minimize.c) ----
ICPGPU_HDI inline int solve_quadratic(double a, double b, double c, double& r0, double& r1) {
  if (a == 0) {
    if (b == 0) return 0;
    r0 = -c / b;
    return 1;
  }
  const double disc = b * b - 4 * a * c;
  if (disc > 0) {
    if (b == 0) {
      const double r = sqrt(-c / a);
      r0 = -r;
      r1 = r;
    } else {
      const double tmp = -0.5 * (b + (b > 0 ? 1.0 : -1.0) * sqrt(disc));
      const double a1 = tmp / a, a2 = c / tmp;
      r0 = dmin(a1, a2);
      r1 = dmax(a1, a2);
    }
    return 2;
  }
  if (disc == 0) {
    r0 = r1 = -0.5 * b / a;
    return 2;
  }
  return 0;
}

ICPGPU_HDI inline double quad_min(double f0, double fp0, double f1, double zl, double zh) {
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

ICPGPU_HDI inline double cubic_min(double f0, double fp0, double f1, double fp1, double zl, double zh) {
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

ICPGPU_HDI inline double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax) {
  double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
  if (ymin > ymax) {
    const double tmp = ymin;
    ymin = ymax;
    ymax = tmp;
  }
  const double y = (fpb != fpb) ? quad_min(fa, fpa * (b - a), fb, ymin, ymax)
                                : cubic_min(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax);  // order 3
  return a + y * (b - a);
}

// the objective along the current search line, with GSL's value caches
template <class EvalFn>
class LineFunction {
 public:
  ICPGPU_HDI explicit LineFunction(EvalFn& eval) : eval_(eval) {}
  ICPGPU_HDI void reset(const V6& x, double f, const V6& g, const V6& p) {
    x0_ = x;
    p_ = p;
    x_alpha_ = x;
    g_alpha_ = g;
    f_alpha_ = f;
    df_alpha_ = dot(g, p);
    f_key_ = df_key_ = x_key_ = g_key_ = 0.0;
  }
  ICPGPU_HDI bool ok() const { return ok_; }
  ICPGPU_HDI double f(double alpha) {
    if (alpha == f_key_) return f_alpha_;
    move(alpha);
    // One device reduction yields the value AND the gradient (same sums), so the gradient is cached here as well: GSL
    // asks for df(alpha) right after f(alpha) on every accepted trial point, which would otherwise be a second evaluation
    // producing bit for bit the same numbers.
    Eval e;
    ok_ = ok_ && eval_(x_alpha_, e);
    f_alpha_ = e.f;
    f_key_ = alpha;
    g_alpha_ = e.g;
    g_key_ = alpha;
    return f_alpha_;
  }
  ICPGPU_HDI double df(double alpha) {
    if (alpha == df_key_) return df_alpha_;
    move(alpha);
    if (alpha != g_key_) {
      Eval e;
      ok_ = ok_ && eval_(x_alpha_, e);
      g_alpha_ = e.g;
      g_key_ = alpha;
    }
    df_alpha_ = dot(g_alpha_, p_);
    df_key_ = alpha;
    return df_alpha_;
  }
  ICPGPU_HDI void fdf(double alpha, double& f_out, double& df_out) {
    if (alpha == f_key_ || alpha == df_key_) {
      f_out = f(alpha);
      df_out = df(alpha);
      return;
    }
    move(alpha);
    Eval e;
    ok_ = ok_ && eval_(x_alpha_, e);
    f_alpha_ = e.f;
    g_alpha_ = e.g;
    f_key_ = g_key_ = alpha;
    df_alpha_ = dot(g_alpha_, p_);
    df_key_ = alpha;
    f_out = f_alpha_;
    df_out = df_alpha_;
  }
  ICPGPU_HDI const V6& x_alpha() const { return x_alpha_; }
  ICPGPU_HDI const V6& g_alpha() const { return g_alpha_; }

 private:
  ICPGPU_HDI void move(double alpha) {
    if (alpha == x_key_) return;
    _Pragma("unroll")
    for (int i = 0; i < 6; ++i) x_alpha_[i] = x0_[i] + alpha * p_[i];
    x_key_ = alpha;
  }
  EvalFn& eval_;
  V6 x0_{}, p_{}, x_alpha_{}, g_alpha_{};
  double f_alpha_ = 0, df_alpha_ = 0, f_key_ = 0, df_key_ = 0, x_key_ = 0, g_key_ = 0;
  bool ok_ = true;
};

enum class Line { Found, NoProgress };

template <class EvalFn>
ICPGPU_HDI Line line_search(LineFunction<EvalFn>& fn, double alpha1, double& alpha_out) {
  constexpr double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5;
  constexpr int bracket_iters = 100, section_iters = 100;
  const double kNaN = __builtin_nan("");
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
    if (fabs(fpalpha) <= -sigma * fp0) {
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
      if (fabs(fpalpha) <= -sigma * fp0) {
        alpha_out = alpha;
        return Line::Found;
      }
      if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
        b = a; fb = fa; fpb = fpa;
      }
      a = alpha; fa = falpha; fpa = fpalpha;
    }
  }
  (void)fb;
  return Line::Found;  // iteration budget exhausted: alpha_out keeps the caller's 0.0, as in GSL
}

// runs <= max_inner BFGS steps from x (gradient tolerance gradient_tol, PCL's line-search constants); x is updated in place.
// at_x (nullable): value and gradient at the start point if the caller already has them (saves one evaluation)
template <class EvalFn>
ICPGPU_HDI Status minimize(EvalFn& eval, V6& x, int max_inner, double gradient_tol, const Eval* at_x) {
  Eval e0;
  if (at_x) e0 = *at_x;  // the caller has just evaluated value and gradient at x
  else if (!eval(x, e0)) return kDeviceError;
  double f = e0.f;
  V6 g = e0.g, x0 = x, g0 = g, p;
  double g0norm = norm(g0);
  _Pragma("unroll")
  for (int i = 0; i < 6; ++i) p[i] = -g[i] / g0norm;
  double pnorm = norm(p), fp0 = -g0norm, delta_f = 0.0;
  LineFunction<EvalFn> line(eval);
  line.reset(x0, f, g0, p);

  int inner = 0;
  bool no_progress = false, success = false;
  do {
    ++inner;
    // ---- one vector_bfgs2 iteration ----
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0 || g0norm != g0norm) {
      no_progress = true;
      break;
    }
    const double f_before = f;
    double alpha1 = 1.0;  // |step_size|
    if (delta_f < 0) alpha1 = dmin(1.0, 2.0 * dmax(-delta_f, 10 * DBL_EPSILON * fabs(f_before)) / (-fp0));
    double alpha = 0.0;
    const Line ls = line_search(line, alpha1, alpha);
    if (!line.ok()) return kDeviceError;
    if (ls == Line::NoProgress) {
      no_progress = true;
      break;
    }
    double df_unused;
    line.fdf(alpha, f, df_unused);
    if (!line.ok()) return kDeviceError;
    x = line.x_alpha();
    g = line.g_alpha();
    delta_f = f - f_before;
    // memoryless BFGS direction p' = g - A dx - B dg
    V6 dx, dg;
    _Pragma("unroll")
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
    _Pragma("unroll")
    for (int i = 0; i < 6; ++i) p[i] = g[i] - A * dx[i] - B * dg[i];
    g0 = g;
    x0 = x;
    g0norm = norm(g0);
    pnorm = norm(p);
    const double dir = dot(p, g) >= 0.0 ? -1.0 : 1.0;
    _Pragma("unroll")
    for (int i = 0; i < 6; ++i) p[i] *= dir / pnorm;
    pnorm = norm(p);
    fp0 = dot(p, g0);
    line.reset(x0, f, g0, p);
    // ---- PCL: testGradient ----
    success = norm(g) < gradient_tol;
  } while (!success && inner < max_inner);
  if (no_progress || success || inner == max_inner) return kOk;
  return kDidNotConverge;
}

}  // namespace gicp
}  // namespace icpgpu
