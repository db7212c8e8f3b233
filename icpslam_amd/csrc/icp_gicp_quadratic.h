// icp_gicp_quadratic.h -- GICP's inner objective as an exact QUADRATIC FORM, evaluated on the host (INTERNAL, host only).
//
// Inside one outer iteration of pcl::GeneralizedIterativeClosestPoint (reached from /root/reference/src/icpslam/icp_odometer.cpp:198
// and src/icpslam/octree_mapper.cpp:114) the correspondences and their Mahalanobis matrices M_i are FIXED; BFGS then evaluates
//     f(x) = 1/m sum_i r_i^T M_i r_i ,   r_i = T(x) p_i - q_i ,   T(x) = applyState(base, x)  (a float 4x4)
// ~35 times, every evaluation a pass over all correspondences (on the GPU: a dependent host <-> device round trip of ~6 us, 2/3 of
// a scan of the reference's pipeline).  Taken over the reals, r_i is LINEAR in the twelve entries of T, so f and the twelve raw
// gradient sums are polynomials of degree <= 2 in those entries whose coefficients do not depend on x:
//     A[e][e'][c][d] = sum_i p~_e p~_e' (M_i)_cd      p~ = (p.x, p.y, p.z, 1)          60 distinct sums
//     Bq[e][c]       = sum_i p~_e (M_i q_i)_c                                            12
//     cq             = sum_i q_i^T M_i q_i                                                1
// ONE device pass per outer iteration collects these 73 sums (gicp_quadratic_kernel, icp_gicp.hip: exact double-double sums of
// float64 terms, as the evaluation kernels keep theirs) plus m and sum d2; every BFGS evaluation is then ~2 k flops on the host:
//     G[e][c]  = sum_i p~_e (M_i r_i)_c = sum_{d,e'} T[d][e'] A[e][e'][c][d] - Bq[e][c]
//     sum M r            = G[3][.]
//     sum (base p)(M r)^T = B G                       (B = base as a 3x4 matrix; PCL's rotation gradient uses base p)
//     sum r^T M r        = sum_{c,e} T[c][e] (G[e][c] - Bq[e][c]) + cq
// The differences of large numbers in these expressions (|p|^2 / |r|^2 ~ 1e6 and more) are taken in double-double arithmetic.
//
// What this is NOT: PCL applies the float matrix T to every point in FLOAT32, which adds a rounding noise of ~3e-6 m per point to
// r_i; the quadratic form is the objective without that noise (the oracle restates it as GICP_SUMS_SMOOTH).  BFGS and the outer
// loop's 1e-6 m stop amplify any perturbation, so the registration lands within the BASELINE tolerance of the PCL-ordered
// evaluation on most pairs, not on its bits -- about as far from it as PCL's own evaluation moves when its sums are re-ordered
// (profiles/r05_gicp_quadratic.txt).  Hence an OPT-IN mode (icpgpu_params.gicp_inner = ICPGPU_GICP_INNER_QUADRATIC); the default
// stays the per-point evaluation, bit-identical to the oracle.
#pragma once

#include <cstring>

namespace icpgpu {
namespace gicp {

constexpr int kQuadSums = 75;  // 60 A, 12 Bq, cq, m, sum d2 (icp_kernels.h: kGicpQuadSums)

struct QuadForm {
  // A[e][e'][c][d]: the double-double sum as (h, l, lo): hi = h + l split at 26 bits (h * a float is exact), lo as it came
  double Ah[4][4][3][3], Al[4][4][3][3], Alo[4][4][3][3];
  double Bq_hi[4][3], Bq_lo[4][3];
  double cq_hi, cq_lo;
  double m, d2;
  double B[3][4];  // base, row-major 3x4, the float entries as doubles
  // the same A, laid out for quad_form_sums: [k = e' * 3 + d][j = e * 3 + c] -- step k of the twelve accumulators G[e][c] reads
  // twelve consecutive numbers (round 6: the accumulators advance together as vectors)
  alignas(32) double Vh[12][12], Vl[12][12], Vlo[12][12];
  alignas(32) double nBq_hi[12], nBq_lo[12];  // -Bq[e][c] at j = e * 3 + c
};

namespace quad_detail {
inline void two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}
inline void split26(double a, double& h, double& l) {  // Veltkamp: h holds the upper 26 bits of a, l the rest (both exact)
  const double t = 134217729.0 * a;                    // 2^27 + 1
  h = t - (t - a);
  l = a - h;
}
struct DD {
  double hi, lo;
};
// acc += (h + l + lo) * t for a t with <= 24 significant bits: h * t and l * t are exact products
inline void dd_fma_split(DD& acc, double h, double l, double lo, double t) {
  double s, e;
  two_sum(acc.hi, h * t, s, e);
  acc.hi = s;
  acc.lo += e + (l * t + lo * t);
}
inline void dd_add(DD& acc, double hi, double lo) {
  double s, e;
  two_sum(acc.hi, hi, s, e);
  acc.hi = s;
  acc.lo += e + lo;
}
inline void dd_renorm(DD& a) {
  double s, e;
  two_sum(a.hi, a.lo, s, e);
  a.hi = s;
  a.lo = e;
}
constexpr int kPair[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};  // (e, e') -> pair number, e <= e' order
constexpr int kTri[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};                            // (c, d) -> upper-triangle entry
}  // namespace quad_detail

// sums: kQuadSums double-double numbers as (hi, lo) pairs -- [pair * 6 + tri] (60), [60 + e * 3 + c] (12), [72] cq, [73] m, [74] d2
inline void quad_form_load(QuadForm& Q, const double* sums, const float base16[16]) {
  using namespace quad_detail;
  for (int e = 0; e < 4; ++e)
    for (int f = 0; f < 4; ++f)
      for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) {
          const int idx = kPair[e][f] * 6 + kTri[c][d];
          split26(sums[2 * idx], Q.Ah[e][f][c][d], Q.Al[e][f][c][d]);
          Q.Alo[e][f][c][d] = sums[2 * idx + 1];
        }
  for (int e = 0; e < 4; ++e)
    for (int c = 0; c < 3; ++c) {
      Q.Bq_hi[e][c] = sums[2 * (60 + e * 3 + c)];
      Q.Bq_lo[e][c] = sums[2 * (60 + e * 3 + c) + 1];
    }
  for (int e = 0; e < 4; ++e)
    for (int c = 0; c < 3; ++c) {
      for (int f = 0; f < 4; ++f)
        for (int d = 0; d < 3; ++d) {
          Q.Vh[f * 3 + d][e * 3 + c] = Q.Ah[e][f][c][d];
          Q.Vl[f * 3 + d][e * 3 + c] = Q.Al[e][f][c][d];
          Q.Vlo[f * 3 + d][e * 3 + c] = Q.Alo[e][f][c][d];
        }
      Q.nBq_hi[e * 3 + c] = -Q.Bq_hi[e][c];
      Q.nBq_lo[e * 3 + c] = -Q.Bq_lo[e][c];
    }
  Q.cq_hi = sums[2 * 72];
  Q.cq_lo = sums[2 * 72 + 1];
  Q.m = sums[2 * 73] + sums[2 * 73 + 1];
  Q.d2 = sums[2 * 74] + sums[2 * 74 + 1];
  for (int r = 0; r < 3; ++r)
    for (int e = 0; e < 4; ++e) Q.B[r][e] = (double)base16[e * 4 + r];  // column-major float 4x4
}

// The 15 numbers eval_from_sums consumes (icp_gicp_solver_impl.h), at the float matrix T16 (column-major, after apply_state):
// s[0] = m, s[1] = sum r^T M r, s[2..4] = sum M r, s[5..13] = sum (base p)(M r)^T (row-major), s[14] = sum d2
__attribute__((always_inline)) inline void quad_form_sums(const QuadForm& Q, const float T16[16], double* s) {
  using namespace quad_detail;
  double T[3][4];
  for (int d = 0; d < 3; ++d)
    for (int e = 0; e < 4; ++e) T[d][e] = (double)T16[e * 4 + d];
  DD G[4][3];  // G[e][c] - Bq[e][c] folded in at the end
  {
    // The twelve accumulators take the SAME twelve steps (k = e' * 3 + d, in that order) on different numbers: they advance
    // together as three 4-wide vectors -- element-wise IEEE operations, no contraction: the bits of the scalar loop
    // (dd_fma_split, dd_add, dd_renorm above, operation for operation), a third of its instructions and its dependent chains.
    typedef double v4 __attribute__((vector_size(32)));
    v4 hi[3], lo[3];
    for (int v = 0; v < 3; ++v) hi[v] = lo[v] = v4{0.0, 0.0, 0.0, 0.0};
    for (int f = 0; f < 4; ++f)
      for (int d = 0; d < 3; ++d) {
        const int k = f * 3 + d;
        const double ts = T[d][f];
        const v4 t = {ts, ts, ts, ts};
        for (int v = 0; v < 3; ++v) {
          const v4 h = *reinterpret_cast<const v4*>(&Q.Vh[k][4 * v]), l = *reinterpret_cast<const v4*>(&Q.Vl[k][4 * v]),
                   ll = *reinterpret_cast<const v4*>(&Q.Vlo[k][4 * v]);
          const v4 b = h * t;                    // dd_fma_split: two_sum(acc.hi, h * t)
          const v4 s2 = hi[v] + b;
          const v4 bb = s2 - hi[v];
          const v4 e = (hi[v] - (s2 - bb)) + (b - bb);
          hi[v] = s2;
          lo[v] += e + (l * t + ll * t);
        }
      }
    for (int v = 0; v < 3; ++v) {
      const v4 bh = *reinterpret_cast<const v4*>(&Q.nBq_hi[4 * v]), bl = *reinterpret_cast<const v4*>(&Q.nBq_lo[4 * v]);
      v4 s2 = hi[v] + bh;                        // dd_add(acc, -Bq_hi, -Bq_lo)
      v4 bb = s2 - hi[v];
      v4 e = (hi[v] - (s2 - bb)) + (bh - bb);
      hi[v] = s2;
      lo[v] += e + bl;
      s2 = hi[v] + lo[v];                        // dd_renorm
      bb = s2 - hi[v];
      e = (hi[v] - (s2 - bb)) + (lo[v] - bb);
      hi[v] = s2;
      lo[v] = e;
      for (int u = 0; u < 4; ++u) {
        const int j = 4 * v + u;
        G[j / 3][j % 3] = DD{hi[v][u], lo[v][u]};
      }
    }
  }
  s[0] = Q.m;
  s[14] = Q.d2;
  for (int c = 0; c < 3; ++c) s[2 + c] = G[3][c].hi + G[3][c].lo;
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) {
      DD acc{0.0, 0.0};
      for (int e = 0; e < 4; ++e) {
        double h, l;
        split26(G[e][c].hi, h, l);
        dd_fma_split(acc, h, l, G[e][c].lo, Q.B[a][e]);
      }
      s[5 + 3 * a + c] = acc.hi + acc.lo;
    }
  DD f{0.0, 0.0};
  for (int c = 0; c < 3; ++c)
    for (int e = 0; e < 4; ++e) {
      DD w = G[e][c];
      dd_add(w, -Q.Bq_hi[e][c], -Q.Bq_lo[e][c]);
      dd_renorm(w);
      double h, l;
      split26(w.hi, h, l);
      dd_fma_split(f, h, l, w.lo, T[c][e]);
    }
  dd_add(f, Q.cq_hi, Q.cq_lo);
  s[1] = f.hi + f.lo;
}

}  // namespace gicp
}  // namespace icpgpu
