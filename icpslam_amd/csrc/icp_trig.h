// icp_trig.h -- sin and cos of the GICP state maps, the same bits on the host and on the device (INTERNAL).
//
// Why: GICP's inner BFGS (pcl::GeneralizedIterativeClosestPoint::estimateRigidTransformationBFGS, reached from
// /root/reference/src/icpslam/icp_odometer.cpp:198) is a chaotic consumer of its inputs -- an ulp flips a line-search decision
// -- and since round 4 it runs INSIDE a kernel (icp_gicp.hip: gicp_solve_kernel) as well as on the host (the fallback path).
// PCL takes its sines and cosines from the platform's libm (Eigen::AngleAxisf: std::cos/std::sin(float); computeRDerivative:
// std::cos/std::sin(double)); libm results are not portable: glibc 2.35's sin / cos / sinf / cosf differ from the correctly
// rounded value on 0.2 % / 0.1 % / 1.8 % / 0.7 % of random arguments (measured, EXPERIMENTS.md section 9-f1), and the device's
// math library differs from both.  The arithmetic contract therefore says: these four are the CORRECTLY
// ROUNDED functions.  The oracle computes them in binary128 (libquadmath) and rounds once; this file computes them in
// double-double arithmetic:
//   x = k * (pi / 256) + t, |t| <= pi / 512   (Cody-Waite reduction: pi / 256 in three parts, k * part exact for |k| < 2^20)
//   sin t = t + t^3 P(t^2), cos t = 1 - (t^2 / 2 - t^4 Q(t^2))   (the small corrections in float64: they are < 2^-15 of the result)
//   sin x, cos x from sin / cos (i pi / 256) (table, double-double) by the addition theorems, in double-double
// with a relative error below 2^-68 before the one final rounding: the result is the correctly rounded one unless the exact
// value lies within ~2^-20 ulp of a rounding boundary (measured against binary128: 8 one-ulp mismatches in 8 * 10^7 double calls,
// none in 8 * 10^7 float calls; tests/test_trig.py repeats it on 4 * 10^5 arguments).  Every operation is an IEEE float64 operation or an explicit fma -- the library is
// compiled with -ffp-contract=off -- so the host compiler and the device compiler produce the same bits.
// Arguments beyond |x| > 2^19 (never a rotation angle) go through the platform's functions.
#pragma once

#include <cmath>

#include "icp_trig_table.h"

#if defined(__HIPCC__)
#define ICPGPU_HDI __host__ __device__ __attribute__((always_inline))  // (a real call on the device means arguments in scratch memory)
#else
#define ICPGPU_HDI
#endif

namespace icpgpu {
namespace trig {

struct DDv {
  double hi, lo;
};

ICPGPU_HDI inline DDv two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return DDv{s, (a - (s - bb)) + (b - bb)};
}
ICPGPU_HDI inline DDv two_prod(double a, double b) {
  const double p = a * b;
  return DDv{p, __builtin_fma(a, b, -p)};
}
// (a.hi + a.lo) * (b.hi + b.lo), relative error ~2^-104
ICPGPU_HDI inline DDv dd_mul(const DDv& a, const DDv& b) {
  DDv p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return two_sum(p.hi, p.lo);
}
ICPGPU_HDI inline DDv dd_add(const DDv& a, const DDv& b) {
  DDv s = two_sum(a.hi, b.hi);
  s.lo += a.lo + b.lo;
  return two_sum(s.hi, s.lo);
}

// sin(x) and cos(x), correctly rounded (see above)
// `table`: the 128 x 4 doubles of ICPGPU_TRIG_TABLE (a kernel hands over its copy in LDS: a lookup is then ~64 cycles instead of
// a trip to the constant cache / L2)
ICPGPU_HDI inline void sincos_cr_with(const double (*kTable)[4], double x, double* s_out, double* c_out);
ICPGPU_HDI inline const double (*trig_table())[4] {
  static constexpr double kTable[128][4] = {ICPGPU_TRIG_TABLE};
  return kTable;
}
ICPGPU_HDI inline void sincos_cr(double x, double* s_out, double* c_out) { sincos_cr_with(trig_table(), x, s_out, c_out); }
ICPGPU_HDI inline void sincos_cr_with(const double (*kTable)[4], double x, double* s_out, double* c_out) {
  if (!(x >= -524288.0 && x <= 524288.0)) {  // (NaN lands here too)
    *s_out = sin(x);
    *c_out = cos(x);
    return;
  }
  const double kd = __builtin_rint(x * kInvStep);
  const int k = (int)kd;
  // t = x - k * pi / 256 as a double-double: the first product is exact, so is the first difference
  const double t1 = x - kd * kStep1;
  DDv t = two_sum(t1, -(kd * kStep2));
  t.lo -= kd * kStep3;
  t = two_sum(t.hi, t.lo);
  // sin t, cos t
  const DDv u = two_prod(t.hi, t.hi);
  const double u1 = u.hi;
  const double ps = -0x1.5555555555555p-3 + u1 * (0x1.1111111111111p-7 + u1 * (-0x1.a01a01a01a01ap-13 + u1 * 0x1.71de3a556c734p-19));
  const DDv sin_t = two_sum(t.hi, t.lo + t.hi * u1 * ps);
  const double pc = 0x1.5555555555555p-5 + u1 * (-0x1.6c16c16c16c17p-10 + u1 * 0x1.a01a01a01a01ap-16);  // 1/24 - u/720 + u^2/40320
  // h = t^2 / 2 - t^4 (1/24 - ...): cos t = 1 - h
  const double h_lo = 0.5 * u.lo + t.hi * t.lo - u1 * u1 * pc;
  DDv cos_t = two_sum(1.0, -0.5 * u1);
  cos_t.lo -= h_lo;
  cos_t = two_sum(cos_t.hi, cos_t.lo);
  // table entry and quadrant: angle j * pi / 256, j = k mod 512
  const int j = k & 511, q = j >> 7, i = j & 127;
  const DDv S{kTable[i][0], kTable[i][1]}, C{kTable[i][2], kTable[i][3]};
  // sin(a + t) = S cos t + C sin t, cos(a + t) = C cos t - S sin t for a = i pi / 256
  const DDv sa = dd_add(dd_mul(S, cos_t), dd_mul(C, sin_t));
  const DDv neg_sin_t{-sin_t.hi, -sin_t.lo};
  const DDv ca = dd_add(dd_mul(C, cos_t), dd_mul(S, neg_sin_t));
  const double sv = sa.hi + sa.lo, cv = ca.hi + ca.lo;
  // quadrant q: sin(a + q pi / 2), cos(a + q pi / 2)
  switch (q) {
    case 0: *s_out = sv; *c_out = cv; break;
    case 1: *s_out = cv; *c_out = -sv; break;
    case 2: *s_out = -sv; *c_out = -cv; break;
    default: *s_out = -cv; *c_out = sv; break;
  }
}

// float versions: the correctly rounded float of sin / cos of a float argument
ICPGPU_HDI inline void sincosf_cr(float x, float* s_out, float* c_out) {
  double s, c;
  sincos_cr((double)x, &s, &c);
  *s_out = (float)s;
  *c_out = (float)c;
}

}  // namespace trig
}  // namespace icpgpu
