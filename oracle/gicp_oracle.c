/*
 * gicp_oracle.c -- CPU ORACLE for the GICP mode.  TEST INFRASTRUCTURE ONLY.
 *
 *      ***  PARITY UNPINNED  ***   (see icp_oracle.h)
 *
 * Restates pcl::GeneralizedIterativeClosestPoint<PointXYZ,PointXYZ> of PCL 1.8.x -- the class the reference literally
 * instantiates at /root/reference/src/icpslam/icp_odometer.cpp:188 and src/icpslam/octree_mapper.cpp:104 with
 * max iterations 10 / 30, transformation epsilon 1e-6, correspondence distance 1.0 (icp_odometer.h:63-65):
 *   computeCovariances   : 20 nearest neighbours (incl. the point), covariance in double from float products,
 *                          SVD, singular values replaced by (1, 1, gicp_epsilon = 1e-3)
 *   computeTransformation: per outer iteration 1-NN of (transformation * guess * p) in the target, keep d2 < r^2,
 *                          M_i = (C_t[j] + R C_s[i] R^T)^-1, minimise (1/m) sum r^T M r over x = (t, roll, pitch, yaw)
 *                          with PCL's BFGS (a port of GSL's vector_bfgs2 + Fletcher line search), <= 20 inner steps,
 *                          gradient tolerance 1e-2; stop when nr >= max_iterations or delta < 1
 *                          (rotation entries scaled by 1/rotation_epsilon = 1/2e-3, the rest by 1/transformation_epsilon)
 * PCL is not in /root/reference; the constants above are PCL's constructor defaults the reference leaves untouched.
 *
 * Arithmetic choices shared with the HIP implementation: point transforms use the fmaf chain of
 * icp_oracle.c; NN keys are (d2, lowest index); the 20-NN set is the 20 smallest (d2, index) keys; the cost and gradient
 * sums of a BFGS evaluation are the EXACT sums of their float64 terms, rounded once (eval_sums); the regularised covariance
 * and the Mahalanobis inverse are stored symmetric (one triangle mirrored).  The last two differ from PCL's own
 * evaluation by no more than its rounding error and are what lets a parallel implementation be compared bit for bit.
 */
#include <float.h>
#include <math.h>
#include <quadmath.h>
#include <stdlib.h>
#include <string.h>

#include "icp_oracle.h"
#include "oracle_internal.h"

#define GICP_K 20
#define GICP_EPSILON 1e-3
#define GICP_ROTATION_EPSILON 2e-3
#define GICP_MAX_INNER 20
#define GICP_GRADIENT_TOL 1e-2

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */
static inline void xform_point(const float T[16], const float* s, float* p) {
  p[0] = fmaf(T[8], s[2], fmaf(T[4], s[1], fmaf(T[0], s[0], T[12])));
  p[1] = fmaf(T[9], s[2], fmaf(T[5], s[1], fmaf(T[1], s[0], T[13])));
  p[2] = fmaf(T[10], s[2], fmaf(T[6], s[1], fmaf(T[2], s[0], T[14])));
}

static void mat4f_mul(const float A[16], const float B[16], float C[16]) { /* column-major, float accumulate */
  float R[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
      R[c * 4 + r] = s;
    }
  memcpy(C, R, sizeof(R));
}

static void mat4f_identity(float M[16]) {
  memset(M, 0, 16 * sizeof(float));
  M[0] = M[5] = M[10] = M[15] = 1.0f;
}

/* 3x3 symmetric-positive inverse via adjugate (Eigen's fixed-size 3x3 inverse), row-major */
static void inv3(const double A[9], double Inv[9]) {
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = 1.0 / det;
  Inv[0] = c00 * id;
  Inv[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  Inv[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Inv[3] = c01 * id;
  Inv[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  Inv[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Inv[6] = c02 * id;
  Inv[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  Inv[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* ------------------------------------------------------------------------------------------ */
/* Eigen::JacobiSVD<Matrix3d>(cov, ComputeFullU).matrixU()                                     */
/* ------------------------------------------------------------------------------------------ */
/* What computeCovariances calls (PCL registration/impl/gicp.hpp: `Eigen::JacobiSVD<Eigen::Matrix3d> svd(cov,
 * Eigen::ComputeFullU); ... col = svd.matrixU().col(k)`), restated from Eigen 3.3's SVD/JacobiSVD.h and Jacobi/Jacobi.h (not
 * under /root/reference; unpinned like PCL itself): the TWO-SIDED Jacobi iteration on the matrix scaled by its largest
 * entry -- pairs (1,0), (2,0), (2,1); a pair is rotated while |w_pq| or |w_qp| exceeds max(DBL_MIN, 2 eps * maxDiagEntry),
 * maxDiagEntry the largest |diagonal| met so far (the stopping rule is relative to the LARGEST singular value: a near-planar
 * patch does not make it hover, unlike the column-relative rule of the one-sided iteration this replaced in round 3);
 * real_2x2_jacobi_svd = a rotation that symmetrises the 2x2 block followed by makeJacobi's symmetric Jacobi rotation; then
 * signs of the diagonal into U's columns and a descending selection sort of the singular values.  Row-major 3x3 arrays.
 * The sweep cap (Eigen has none) only guards non-finite input; the GPU kernel (icp_gicp.hip, svd3_eigen_u) performs the same
 * operations in the same order. */
#define ORC_SVD_MAX_SWEEPS 64
static void rot_rows(double* W, int p, int q, double c, double s) { /* applyOnTheLeft(p, q, {c, s}) */
  if (c == 1.0 && s == 0.0) return;
  for (int i = 0; i < 3; ++i) {
    const double xi = W[3 * p + i], yi = W[3 * q + i];
    W[3 * p + i] = c * xi + s * yi;
    W[3 * q + i] = -s * xi + c * yi;
  }
}
static void rot_cols(double* W, int p, int q, double c, double s) { /* applyOnTheRight(p, q, {c, s}): the transposed rotation on columns */
  if (c == 1.0 && s == 0.0) return;
  for (int i = 0; i < 3; ++i) {
    const double xi = W[3 * i + p], yi = W[3 * i + q];
    W[3 * i + p] = c * xi - s * yi;
    W[3 * i + q] = s * xi + c * yi;
  }
}
void orc_svd3_eigen_u(const double A[9], double U[9], double sv[3]) {
  double scale = 0.0;
  for (int i = 0; i < 9; ++i)
    if (fabs(A[i]) > scale) scale = fabs(A[i]);
  if (scale == 0.0) scale = 1.0;
  double W[9];
  for (int i = 0; i < 9; ++i) W[i] = A[i] / scale;
  for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0;
  const double precision = 2.0 * DBL_EPSILON, consider_as_zero = DBL_MIN;
  double max_diag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
  for (int sweep = 0; sweep < ORC_SVD_MAX_SWEEPS; ++sweep) {
    int finished = 1;
    for (int p = 1; p < 3; ++p)
      for (int q = 0; q < p; ++q) {
        const double threshold = fmax(consider_as_zero, precision * max_diag);
        if (!(fabs(W[3 * p + q]) > threshold || fabs(W[3 * q + p]) > threshold)) continue;
        finished = 0;
        /* real_2x2_jacobi_svd on m = [w_pp w_pq; w_qp w_qq] */
        double m00 = W[3 * p + p], m01 = W[3 * p + q], m10 = W[3 * q + p], m11 = W[3 * q + q];
        double c1, s1;
        const double t = m00 + m11, d = m10 - m01;
        if (fabs(d) < DBL_MIN) {
          s1 = 0.0;
          c1 = 1.0;
        } else {
          const double u = t / d, tmp = sqrt(1.0 + u * u);
          s1 = 1.0 / tmp;
          c1 = u / tmp;
        }
        if (!(c1 == 1.0 && s1 == 0.0)) { /* m.applyOnTheLeft(0, 1, rot1) */
          const double a0 = m00, a1 = m01, b0 = m10, b1 = m11;
          m00 = c1 * a0 + s1 * b0;
          m01 = c1 * a1 + s1 * b1;
          m10 = -s1 * a0 + c1 * b0;
          m11 = -s1 * a1 + c1 * b1;
        }
        /* j_right = makeJacobi(m00, m01, m11) */
        double cr, sr;
        const double deno = 2.0 * fabs(m01);
        if (deno < DBL_MIN) {
          cr = 1.0;
          sr = 0.0;
        } else {
          const double tau = (m00 - m11) / deno, w = sqrt(tau * tau + 1.0);
          const double tt = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
          const double sign_t = tt > 0.0 ? 1.0 : -1.0, n = 1.0 / sqrt(tt * tt + 1.0);
          sr = -sign_t * (m01 / fabs(m01)) * fabs(tt) * n;
          cr = n;
        }
        /* j_left = rot1 * j_right^T:  c = c1 cr - s1 (-sr),  s = c1 (-sr) + s1 cr */
        const double cl = c1 * cr - s1 * (-sr), sl = c1 * (-sr) + s1 * cr;
        rot_rows(W, p, q, cl, sl);           /* m_workMatrix.applyOnTheLeft(p, q, j_left) */
        rot_cols(U, p, q, cl, -sl);          /* m_matrixU.applyOnTheRight(p, q, j_left.transpose()) */
        rot_cols(W, p, q, cr, sr);           /* m_workMatrix.applyOnTheRight(p, q, j_right) */
        max_diag = fmax(max_diag, fmax(fabs(W[3 * p + p]), fabs(W[3 * q + q])));
      }
    if (finished) break;
  }
  for (int i = 0; i < 3; ++i) { /* signs of the diagonal into U; singular values */
    const double a = fabs(W[4 * i]);
    sv[i] = a;
    if (a != 0.0) {
      const double f = W[4 * i] / a;
      for (int r = 0; r < 3; ++r) U[3 * r + i] *= f;
    }
  }
  for (int i = 0; i < 3; ++i) sv[i] *= scale;
  for (int i = 0; i < 3; ++i) { /* descending selection sort, columns of U follow */
    int pos = i;
    for (int k = i + 1; k < 3; ++k)
      if (sv[k] > sv[pos]) pos = k;
    if (sv[pos] == 0.0) break;
    if (pos != i) {
      const double ts = sv[i];
      sv[i] = sv[pos];
      sv[pos] = ts;
      for (int r = 0; r < 3; ++r) {
        const double tu = U[3 * r + i];
        U[3 * r + i] = U[3 * r + pos];
        U[3 * r + pos] = tu;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* computeCovariances                                                                          */
/* ------------------------------------------------------------------------------------------ */
int orc_gicp_covariances_ex(const float* cloud, size_t n, int arith, int pcl_order, double* cov_out /* n x 9 row-major */) {
  if (n < GICP_K) return -1; /* PCL: "Number or points in cloud is less than k_correspondences_" */
  void* tree = orc_kd_build(cloud, n, arith);
  int32_t idx[GICP_K];
  float d2[GICP_K];
  for (size_t i = 0; i < n; ++i) {
    orc_kd_knn(tree, cloud + 4 * i, GICP_K, idx, d2);
    double mean[3] = {0, 0, 0}, cov[9] = {0};
    for (int j = 0; j < GICP_K; ++j) {
      const float* pt = cloud + 4 * (size_t)idx[j];
      mean[0] += pt[0];
      mean[1] += pt[1];
      mean[2] += pt[2];
      /* float * float products (rounded to float), accumulated in double -- as the C++ expression evaluates */
      cov[0] += (double)(pt[0] * pt[0]);
      cov[3] += (double)(pt[1] * pt[0]);
      cov[4] += (double)(pt[1] * pt[1]);
      cov[6] += (double)(pt[2] * pt[0]);
      cov[7] += (double)(pt[2] * pt[1]);
      cov[8] += (double)(pt[2] * pt[2]);
    }
    for (int a = 0; a < 3; ++a) mean[a] /= (double)GICP_K;
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l <= k; ++l) {
        cov[3 * k + l] /= (double)GICP_K;
        cov[3 * k + l] -= mean[k] * mean[l];
        cov[3 * l + k] = cov[3 * k + l];
      }
    double U[9], s[3];
    orc_svd3_eigen_u(cov, U, s);
    /* cov = sum_k v_k u_k u_k^T, v = (1, 1, epsilon).  The lower triangle is computed -- entry (r, c) = sum_k (v_k u_rk) u_ck,
     * the expression Eigen evaluates for `v * col * col.transpose()` -- and mirrored: in PCL the two triangles can differ in
     * the last bit of the epsilon term ((eps a) b vs (eps b) a), 1e-19 absolute; an implementation that stores six entries
     * per point (the GPU path) could never reproduce that, and nothing downstream gives it a meaning. */
    double* C = cov_out + 9 * i;
    if (pcl_order) {
      /* ORC_GICP_SUMS_SEQUENTIAL: PCL's own loop, `cov.setZero(); for k: cov += v * col * col.transpose();` -- all nine
       * entries accumulated over k, the two triangles keeping their last-bit differences */
      for (int e = 0; e < 9; ++e) C[e] = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double v = (k == 2) ? GICP_EPSILON : 1.0;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) C[3 * r + c] += v * U[3 * r + k] * U[3 * c + k];
      }
      continue;
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c <= r; ++c) {
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) {
          const double v = (k == 2) ? GICP_EPSILON : 1.0;
          acc += v * U[3 * r + k] * U[3 * c + k];
        }
        C[3 * r + c] = acc;
        C[3 * c + r] = acc;
      }
  }
  orc_kd_free(tree);
  return 0;
}

int orc_gicp_covariances(const float* cloud, size_t n, int arith, double* cov_out) {
  return orc_gicp_covariances_ex(cloud, n, arith, 0, cov_out);
}

/* ------------------------------------------------------------------------------------------ */
/* applyState / cost / gradient                                                                */
/* ------------------------------------------------------------------------------------------ */
/* The sines and cosines of the state maps.  PCL takes them from the platform's libm (Eigen::AngleAxisf: cosf / sinf;
 * computeRDerivative: cos / sin), whose results are not portable -- glibc 2.35's differ from the correctly rounded value on
 * 0.2 % (sin), 0.1 % (cos), 1.8 % (sinf), 0.7 % (cosf) of random arguments -- and BFGS amplifies an ulp.  So, like the sums:
 *   ORC_GICP_SUMS_EXACT       (the contract a parallel implementation is compared with bit for bit): the CORRECTLY ROUNDED
 *                             functions -- evaluated in binary128 (libquadmath: 113 bits), rounded once;
 *   ORC_GICP_SUMS_SEQUENTIAL* (PCL's own evaluation): this platform's libm, as a PCL built here would call it.
 * (The product computes the correctly rounded values its own way -- double-double arithmetic, icpslam_amd/csrc/icp_trig.h --
 * so agreement is a test of both.)  Thread-local: the mode of the orc_icp_align call running on this thread. */
static __thread int tl_trig_cr = 1;
static double o_sin(double x) { return tl_trig_cr ? (double)sinq((__float128)x) : sin(x); }
static double o_cos(double x) { return tl_trig_cr ? (double)cosq((__float128)x) : cos(x); }
static float o_sinf(float x) { return tl_trig_cr ? (float)sinq((__float128)x) : sinf(x); }
static float o_cosf(float x) { return tl_trig_cr ? (float)cosq((__float128)x) : cosf(x); }
/* t <- Rz(x5) Ry(x4) Rx(x3) * t.R ; t.col(3) += (x0, x1, x2).  float, like Eigen's Matrix4f/AngleAxisf path:
 * each AngleAxis becomes a quaternion (cos a/2, sin a/2 * axis), the product is converted to a rotation matrix. */
static void apply_state(float t[16], const double x[6]) {
  const float hx = 0.5f * (float)x[3], hy = 0.5f * (float)x[4], hz = 0.5f * (float)x[5];
  const float qx[4] = {o_cosf(hx), o_sinf(hx), 0.f, 0.f}; /* w, x, y, z */
  const float qy[4] = {o_cosf(hy), 0.f, o_sinf(hy), 0.f};
  const float qz[4] = {o_cosf(hz), 0.f, 0.f, o_sinf(hz)};
  float a[4], q[4];
  /* a = qz * qy */
  a[0] = qz[0] * qy[0] - qz[1] * qy[1] - qz[2] * qy[2] - qz[3] * qy[3];
  a[1] = qz[0] * qy[1] + qz[1] * qy[0] + qz[2] * qy[3] - qz[3] * qy[2];
  a[2] = qz[0] * qy[2] - qz[1] * qy[3] + qz[2] * qy[0] + qz[3] * qy[1];
  a[3] = qz[0] * qy[3] + qz[1] * qy[2] - qz[2] * qy[1] + qz[3] * qy[0];
  /* q = a * qx */
  q[0] = a[0] * qx[0] - a[1] * qx[1] - a[2] * qx[2] - a[3] * qx[3];
  q[1] = a[0] * qx[1] + a[1] * qx[0] + a[2] * qx[3] - a[3] * qx[2];
  q[2] = a[0] * qx[2] - a[1] * qx[3] + a[2] * qx[0] + a[3] * qx[1];
  q[3] = a[0] * qx[3] + a[1] * qx[2] - a[2] * qx[1] + a[3] * qx[0];
  const float tx = 2.f * q[1], ty = 2.f * q[2], tz = 2.f * q[3];
  const float twx = tx * q[0], twy = ty * q[0], twz = tz * q[0];
  const float txx = tx * q[1], txy = ty * q[1], txz = tz * q[1];
  const float tyy = ty * q[2], tyz = tz * q[2], tzz = tz * q[3];
  float R[9]; /* row-major */
  R[0] = 1.f - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.f - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.f - (txx + tyy);
  float nr[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += R[3 * r + k] * t[c * 4 + k];
      nr[3 * r + c] = s;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) t[c * 4 + r] = nr[3 * r + c];
  t[12] += (float)x[0];
  t[13] += (float)x[1];
  t[14] += (float)x[2];
}

typedef struct {
  const float* src; /* the source cloud (PCL's `output`, a copy of the input) */
  const float* tgt;
  const int32_t* si;
  const int32_t* ti;
  int m;
  const double* maha; /* indexed by SOURCE index, 9 doubles each */
  float base[16];     /* base_transformation_ = guess */
  int sequential;     /* 1: ORC_GICP_SUMS_SEQUENTIAL, PCL's plain float64 loop instead of the exact sums; 2: ..._REVERSED */
  int smooth;         /* 1: ORC_GICP_SUMS_SMOOTH, exact sums of the QUADRATIC objective (see eval_sums) */
} gicp_problem;

/* Order-independent sums.  PCL adds the cost and the gradient terms in plain float64, one after the other; BFGS then
 * consumes them chaotically (an ulp in a sum can flip a line-search decision, and the outer delta < 1 stop sits on a
 * 1e-6 m threshold), so any implementation that adds in another order -- every parallel one -- drifts away from the
 * sequential result on a few percent of the pairs.  The oracle therefore defines each of the 13 sums as the EXACT sum of
 * its float64 terms, rounded once: a three-fold error-free expansion (TwoSum cascade) over the sequential loop.  It differs
 * from PCL's plain sum by the rounding PCL accumulates (~1e-13 relative at 50k terms), and it is the same number whatever
 * the order of summation, which is what makes a bit-for-bit comparison with the GPU path meaningful. */
typedef struct { double hi, mid, lo; } sum3;
static inline void two_sum(double a, double b, double* s, double* e) {
  *s = a + b;
  const double bb = *s - a;
  *e = (a - (*s - bb)) + (b - bb);
}
static inline void sum3_add(sum3* a, double t) {
  double e1, e2;
  two_sum(a->hi, t, &a->hi, &e1);
  two_sum(a->mid, e1, &a->mid, &e2);
  a->lo += e2;
}
static inline double sum3_value(const sum3* a) {
  double s, e;
  two_sum(a->mid, a->lo, &s, &e); /* renormalise from the bottom: hi + (mid + lo), with the low error folded back in */
  double h, l;
  two_sum(a->hi, s, &h, &l);
  return h + (l + e);
}

/* f and the 12 raw gradient sums (g_t(3), R(9)) at x */
static void eval_sums(const gicp_problem* P, const double x[6], double* f, double gt[3], double Rm[9]) {
  float T[16];
  memcpy(T, P->base, sizeof(T));
  apply_state(T, x);
  if (P->sequential) {
    /* PCL's OptimizationFunctorWithIndices::operator() / fdf: `f += double(res.transpose() * temp)`,
     * `g.head<3>() += temp`, `R += p_src3 * temp.transpose()` -- one float64 accumulator each, in correspondence order */
    double acc = 0.0;
    gt[0] = gt[1] = gt[2] = 0.0;
    for (int k = 0; k < 9; ++k) Rm[k] = 0.0;
    for (int ii = 0; ii < P->m; ++ii) {
      const int i = P->sequential == 2 ? P->m - 1 - ii : ii; /* 2: the same loop run backwards (sensitivity probe) */
      const float* ps = P->src + 4 * (size_t)P->si[i];
      const float* pt = P->tgt + 4 * (size_t)P->ti[i];
      float pp[3], pb[3];
      xform_point(T, ps, pp);
      const double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
      const double* M = P->maha + 9 * (size_t)P->si[i];
      double temp[3];
      for (int r = 0; r < 3; ++r) temp[r] = M[3 * r] * res[0] + M[3 * r + 1] * res[1] + M[3 * r + 2] * res[2];
      acc += res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2];
      for (int r = 0; r < 3; ++r) gt[r] += temp[r];
      xform_point(P->base, ps, pb);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rm[3 * r + c] += (double)pb[r] * temp[c];
    }
    *f = acc;
    return;
  }
  sum3 acc = {0, 0, 0}, gts[3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Rms[9];
  for (int k = 0; k < 9; ++k) Rms[k].hi = Rms[k].mid = Rms[k].lo = 0.0;
  if (P->smooth) {
    /* ORC_GICP_SUMS_SMOOTH: the objective WITHOUT the float32 rounding of the transformed points.  PCL applies the float matrix T
     * to every source point in float32 (a 3e-6 m noise per point at 50 m); with T p and base p taken as real numbers (here:
     * float64, from the same float matrix) the cost of an outer iteration is an exact QUADRATIC form in T's twelve entries --
     * what the GPU's quadratic inner solver (icpgpu_params.gicp_inner = 1) evaluates on the host from 73 sums collected once. */
    for (int i = 0; i < P->m; ++i) {
      const float* ps = P->src + 4 * (size_t)P->si[i];
      const float* pt = P->tgt + 4 * (size_t)P->ti[i];
      double pp[3], pb[3];
      for (int r = 0; r < 3; ++r) {
        pp[r] = (double)T[r] * ps[0] + (double)T[4 + r] * ps[1] + (double)T[8 + r] * ps[2] + (double)T[12 + r];
        pb[r] = (double)P->base[r] * ps[0] + (double)P->base[4 + r] * ps[1] + (double)P->base[8 + r] * ps[2] + (double)P->base[12 + r];
      }
      const double res[3] = {pp[0] - (double)pt[0], pp[1] - (double)pt[1], pp[2] - (double)pt[2]};
      const double* M = P->maha + 9 * (size_t)P->si[i];
      double temp[3];
      for (int r = 0; r < 3; ++r) temp[r] = M[3 * r] * res[0] + M[3 * r + 1] * res[1] + M[3 * r + 2] * res[2];
      sum3_add(&acc, res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2]);
      for (int r = 0; r < 3; ++r) sum3_add(&gts[r], temp[r]);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sum3_add(&Rms[3 * r + c], pb[r] * temp[c]);
    }
    *f = sum3_value(&acc);
    for (int r = 0; r < 3; ++r) gt[r] = sum3_value(&gts[r]);
    for (int k = 0; k < 9; ++k) Rm[k] = sum3_value(&Rms[k]);
    return;
  }
  for (int i = 0; i < P->m; ++i) {
    const float* ps = P->src + 4 * (size_t)P->si[i];
    const float* pt = P->tgt + 4 * (size_t)P->ti[i];
    float pp[3], pb[3];
    xform_point(T, ps, pp);
    const double res[3] = {(double)(pp[0] - pt[0]), (double)(pp[1] - pt[1]), (double)(pp[2] - pt[2])};
    const double* M = P->maha + 9 * (size_t)P->si[i];
    double temp[3];
    for (int r = 0; r < 3; ++r) temp[r] = M[3 * r] * res[0] + M[3 * r + 1] * res[1] + M[3 * r + 2] * res[2];
    sum3_add(&acc, res[0] * temp[0] + res[1] * temp[1] + res[2] * temp[2]);
    for (int r = 0; r < 3; ++r) sum3_add(&gts[r], temp[r]);
    xform_point(P->base, ps, pb); /* PCL uses base_transformation_ * p_src for the rotation gradient */
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) sum3_add(&Rms[3 * r + c], (double)pb[r] * temp[c]);
  }
  *f = sum3_value(&acc);
  for (int r = 0; r < 3; ++r) gt[r] = sum3_value(&gts[r]);
  for (int k = 0; k < 9; ++k) Rm[k] = sum3_value(&Rms[k]);
}

static void r_derivative(const double x[6], const double R[9], double g[6]) {
  const double phi = x[3], theta = x[4], psi = x[5];
  const double cphi = o_cos(phi), sphi = o_sin(phi), cth = o_cos(theta), sth = o_sin(theta), cpsi = o_cos(psi), spsi = o_sin(psi);
  double dphi[9], dth[9], dpsi[9]; /* row-major */
  dphi[0] = 0; dphi[3] = 0; dphi[6] = 0;
  dphi[1] = sphi * spsi + cphi * cpsi * sth;
  dphi[4] = -cpsi * sphi + cphi * spsi * sth;
  dphi[7] = cphi * cth;
  dphi[2] = cphi * spsi - cpsi * sphi * sth;
  dphi[5] = -cphi * cpsi - sphi * spsi * sth;
  dphi[8] = -cth * sphi;
  dth[0] = -cpsi * sth; dth[3] = -spsi * sth; dth[6] = -cth;
  dth[1] = cpsi * cth * sphi; dth[4] = cth * sphi * spsi; dth[7] = -sphi * sth;
  dth[2] = cphi * cpsi * cth; dth[5] = cphi * cth * spsi; dth[8] = -cphi * sth;
  dpsi[0] = -cth * spsi; dpsi[3] = cpsi * cth; dpsi[6] = 0;
  dpsi[1] = -cphi * cpsi - sphi * spsi * sth; dpsi[4] = -cphi * spsi + cpsi * sphi * sth; dpsi[7] = 0;
  dpsi[2] = cpsi * sphi - cphi * spsi * sth; dpsi[5] = sphi * spsi + cphi * cpsi * sth; dpsi[8] = 0;
  /* matricesInnerProd(mat1, mat2) = sum_ij mat1(j,i) * mat2(i,j) */
  double a = 0, b = 0, c = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a += dphi[3 * j + i] * R[3 * i + j];
      b += dth[3 * j + i] * R[3 * i + j];
      c += dpsi[3 * j + i] * R[3 * i + j];
    }
  g[3] = a;
  g[4] = b;
  g[5] = c;
}

static double cost_f(const gicp_problem* P, const double x[6]) {
  double f, gt[3], Rm[9];
  eval_sums(P, x, &f, gt, Rm);
  return f / (double)P->m;
}

static void cost_fdf(const gicp_problem* P, const double x[6], double* f, double g[6]) {
  double fs, gt[3], Rm[9];
  eval_sums(P, x, &fs, gt, Rm);
  *f = fs / (double)P->m;
  const double sc = 2.0 / (double)P->m;
  for (int r = 0; r < 3; ++r) g[r] = gt[r] * sc;
  for (int k = 0; k < 9; ++k) Rm[k] *= sc;
  r_derivative(x, Rm, g);
}

/* ------------------------------------------------------------------------------------------ */
/* PCL BFGS (pcl/registration/bfgs.h) = GSL vector_bfgs2 + linear_minimize                      */
/* ------------------------------------------------------------------------------------------ */
enum { BFGS_RUNNING = -1, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 1 };

typedef struct {
  const gicp_problem* P;
  double rho, sigma, tau1, tau2, tau3, step_size;
  int order, bracket_iters, section_iters;
  int iter;
  double f, g0norm, pnorm, delta_f, fp0;
  double x0[6], g0[6], p[6], dx0[6], dg0[6], gradient[6];
  /* line-function wrapper with caches */
  double x_alpha[6], g_alpha[6], f_alpha, df_alpha;
  double f_cache_key, df_cache_key, x_cache_key, g_cache_key;
} bfgs_t;

static double dot6(const double* a, const double* b) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s += a[i] * b[i];
  return s;
}
static double nrm6(const double* a) { return sqrt(dot6(a, a)); }

static void w_moveto(bfgs_t* B, double alpha) {
  if (alpha == B->x_cache_key) return;
  for (int i = 0; i < 6; ++i) B->x_alpha[i] = B->x0[i] + alpha * B->p[i];
  B->x_cache_key = alpha;
}
static double w_slope(const bfgs_t* B) { return dot6(B->g_alpha, B->p); }
static double w_f(bfgs_t* B, double alpha) {
  if (alpha == B->f_cache_key) return B->f_alpha;
  w_moveto(B, alpha);
  B->f_alpha = cost_f(B->P, B->x_alpha);
  B->f_cache_key = alpha;
  return B->f_alpha;
}
static double w_df(bfgs_t* B, double alpha) {
  if (alpha == B->df_cache_key) return B->df_alpha;
  w_moveto(B, alpha);
  if (alpha != B->g_cache_key) {
    double f;
    cost_fdf(B->P, B->x_alpha, &f, B->g_alpha); /* PCL's functor->df; f discarded */
    B->g_cache_key = alpha;
  }
  B->df_alpha = w_slope(B);
  B->df_cache_key = alpha;
  return B->df_alpha;
}
static void w_fdf(bfgs_t* B, double alpha, double* f, double* df) {
  if (alpha == B->f_cache_key && alpha == B->df_cache_key) {
    *f = B->f_alpha;
    *df = B->df_alpha;
    return;
  }
  if (alpha == B->f_cache_key || alpha == B->df_cache_key) {
    *f = w_f(B, alpha);
    *df = w_df(B, alpha);
    return;
  }
  w_moveto(B, alpha);
  cost_fdf(B->P, B->x_alpha, &B->f_alpha, B->g_alpha);
  B->f_cache_key = alpha;
  B->g_cache_key = alpha;
  B->df_alpha = w_slope(B);
  B->df_cache_key = alpha;
  *f = B->f_alpha;
  *df = B->df_alpha;
}

static int poly_solve_quadratic(double a, double b, double c, double* x0, double* x1) {
  if (a == 0) {
    if (b == 0) return 0;
    *x0 = -c / b;
    return 1;
  }
  const double disc = b * b - 4 * a * c;
  if (disc > 0) {
    if (b == 0) {
      const double r = sqrt(-c / a);
      *x0 = -r;
      *x1 = r;
    } else {
      const double sgnb = (b > 0 ? 1 : -1);
      const double temp = -0.5 * (b + sgnb * sqrt(disc));
      const double r1 = temp / a, r2 = c / temp;
      if (r1 < r2) {
        *x0 = r1;
        *x1 = r2;
      } else {
        *x0 = r2;
        *x1 = r1;
      }
    }
    return 2;
  } else if (disc == 0) {
    *x0 = -0.5 * b / a;
    *x1 = -0.5 * b / a;
    return 2;
  }
  return 0;
}

static double interp_quad(double f0, double fp0, double f1, double zl, double zh) {
  const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
  const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
  const double c = 2 * (f1 - f0 - fp0);
  double zmin = zl, fmin = fl;
  if (fh < fmin) {
    zmin = zh;
    fmin = fh;
  }
  if (c > 0) {
    const double z = -fp0 / c;
    if (z > zl && z < zh) {
      const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
      if (f < fmin) {
        zmin = z;
        fmin = f;
      }
    }
  }
  return zmin;
}
static double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }
static void check_extremum(double c0, double c1, double c2, double c3, double z, double* zmin, double* fmin) {
  const double y = cubic(c0, c1, c2, c3, z);
  if (y < *fmin) {
    *zmin = z;
    *fmin = y;
  }
}
static double interp_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {
  const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
  const double xi = fp0 + fp1 - 2 * (f1 - f0);
  const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
  double zmin = zl, fmin = cubic(c0, c1, c2, c3, zl), z0, z1;
  check_extremum(c0, c1, c2, c3, zh, &zmin, &fmin);
  const int n = poly_solve_quadratic(3 * c3, 2 * c2, c1, &z0, &z1);
  if (n == 2) {
    if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
    if (z1 > zl && z1 < zh) check_extremum(c0, c1, c2, c3, z1, &zmin, &fmin);
  } else if (n == 1) {
    if (z0 > zl && z0 < zh) check_extremum(c0, c1, c2, c3, z0, &zmin, &fmin);
  }
  return zmin;
}
static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax,
                          int order) {
  double ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
  if (ymin > ymax) {
    const double tmp = ymin;
    ymin = ymax;
    ymax = tmp;
  }
  double y;
  if (order > 2 && !(fpb != fpb))
    y = interp_cubic(fa, fpa * (b - a), fb, fpb * (b - a), ymin, ymax);
  else
    y = interp_quad(fa, fpa * (b - a), fb, ymin, ymax);
  return a + y * (b - a);
}

static int line_search(bfgs_t* B, double alpha1, double* alpha_new) {
  double f0, fp0, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
  double alpha = alpha1, alpha_prev = 0.0;
  double a = 0.0, b = alpha, fa, fb = 0.0, fpa, fpb = 0.0;
  int i = 0;
  w_fdf(B, 0.0, &f0, &fp0);
  falpha_prev = f0;
  fpalpha_prev = fp0;
  fa = f0;
  fpa = fp0;
  while (i++ < B->bracket_iters) { /* bracketing */
    falpha = w_f(B, alpha);
    if (falpha > f0 + alpha * B->rho * fp0 || falpha >= falpha_prev) {
      a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
      b = alpha; fb = falpha; fpb = NAN;
      break;
    }
    fpalpha = w_df(B, alpha);
    if (fabs(fpalpha) <= -B->sigma * fp0) {
      *alpha_new = alpha;
      return BFGS_SUCCESS;
    }
    if (fpalpha >= 0) {
      a = alpha; fa = falpha; fpa = fpalpha;
      b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
      break;
    }
    delta = alpha - alpha_prev;
    {
      const double lower = alpha + delta, upper = alpha + B->tau1 * delta;
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, lower, upper, B->order);
    }
    alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
    alpha = alpha_next;
  }
  while (i++ < B->section_iters) { /* sectioning */
    delta = b - a;
    {
      const double lower = a + B->tau2 * delta, upper = b - B->tau3 * delta;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, lower, upper, B->order);
    }
    falpha = w_f(B, alpha);
    if ((a - alpha) * fpa <= DBL_EPSILON) return BFGS_NOPROGRESS; /* roundoff prevents progress */
    if (falpha > f0 + B->rho * alpha * fp0 || falpha >= fa) {
      b = alpha; fb = falpha; fpb = NAN;
    } else {
      fpalpha = w_df(B, alpha);
      if (fabs(fpalpha) <= -B->sigma * fp0) {
        *alpha_new = alpha;
        return BFGS_SUCCESS;
      }
      if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
        b = a; fb = fa; fpb = fpa;
        a = alpha; fa = falpha; fpa = fpalpha;
      } else {
        a = alpha; fa = falpha; fpa = fpalpha;
      }
    }
  }
  return BFGS_SUCCESS;
}

static void bfgs_init(bfgs_t* B, const gicp_problem* P, const double x[6]) {
  memset(B, 0, sizeof(*B));
  B->P = P;
  B->rho = 0.01; B->sigma = 0.01; B->tau1 = 9; B->tau2 = 0.05; B->tau3 = 0.5; B->order = 3;
  B->step_size = 1.0; B->bracket_iters = 100; B->section_iters = 100;
  cost_fdf(P, x, &B->f, B->gradient);
  memcpy(B->x0, x, sizeof(B->x0));
  memcpy(B->g0, B->gradient, sizeof(B->g0));
  B->g0norm = nrm6(B->g0);
  for (int i = 0; i < 6; ++i) B->p[i] = B->gradient[i] * (-1.0 / B->g0norm);
  B->pnorm = nrm6(B->p);
  B->fp0 = -B->g0norm;
  memcpy(B->x_alpha, B->x0, sizeof(B->x0));
  memcpy(B->g_alpha, B->g0, sizeof(B->g0));
  B->f_alpha = B->f;
  B->df_alpha = w_slope(B);
  B->f_cache_key = B->df_cache_key = B->x_cache_key = B->g_cache_key = 0.0;
}

static int bfgs_step(bfgs_t* B, double x[6]) {
  double alpha = 0.0, alpha1;
  const double f0 = B->f;
  if (B->pnorm == 0.0 || B->g0norm == 0.0 || B->fp0 == 0) return BFGS_NOPROGRESS;
  if (B->delta_f < 0) {
    const double del = fmax(-B->delta_f, 10 * DBL_EPSILON * fabs(f0));
    alpha1 = fmin(1.0, 2.0 * del / (-B->fp0));
  } else
    alpha1 = fabs(B->step_size);
  const int status = line_search(B, alpha1, &alpha);
  if (status != BFGS_SUCCESS) return status;
  /* update_position */
  {
    double f, df;
    w_fdf(B, alpha, &f, &df);
    B->f = f;
    memcpy(x, B->x_alpha, 6 * sizeof(double));
    memcpy(B->gradient, B->g_alpha, 6 * sizeof(double));
  }
  B->delta_f = B->f - f0;
  /* memoryless BFGS direction: p' = g1 - A dx - B dg */
  for (int i = 0; i < 6; ++i) {
    B->dx0[i] = x[i] - B->x0[i];
    B->dg0[i] = B->gradient[i] - B->g0[i];
  }
  const double dxg = dot6(B->dx0, B->gradient), dgg = dot6(B->dg0, B->gradient), dxdg = dot6(B->dx0, B->dg0);
  const double dgnorm = nrm6(B->dg0);
  double A = 0, Bc = 0;
  if (dxdg != 0) {
    Bc = dxg / dxdg;
    A = -(1.0 + dgnorm * dgnorm / dxdg) * Bc + dgg / dxdg;
  }
  for (int i = 0; i < 6; ++i) B->p[i] = B->gradient[i] - A * B->dx0[i] - Bc * B->dg0[i];
  memcpy(B->g0, B->gradient, sizeof(B->g0));
  memcpy(B->x0, x, sizeof(B->x0));
  B->g0norm = nrm6(B->g0);
  B->pnorm = nrm6(B->p);
  const double pg = dot6(B->p, B->gradient);
  const double dir = (pg >= 0.0) ? -1.0 : +1.0;
  for (int i = 0; i < 6; ++i) B->p[i] *= dir / B->pnorm;
  B->pnorm = nrm6(B->p);
  B->fp0 = dot6(B->p, B->g0);
  /* change_direction */
  B->df_alpha = w_slope(B);
  B->df_cache_key = 0.0;
  /* the caches now refer to alpha = 0 of the new line: x_alpha = x0, f_alpha = f, g_alpha = g0 */
  B->f_cache_key = 0.0;
  B->x_cache_key = 0.0;
  B->g_cache_key = 0.0;
  return BFGS_SUCCESS;
}

/* estimateRigidTransformationBFGS: returns 0 ok, -1 not enough points, -2 solver did not converge */
static int estimate_bfgs(const gicp_problem* P, float transformation[16]) {
  if (P->m < 4) return -1;
  double x[6];
  x[0] = transformation[12];
  x[1] = transformation[13];
  x[2] = transformation[14];
  x[3] = atan2((double)transformation[6], (double)transformation[10]);  /* (2,1), (2,2) */
  x[4] = asin(-(double)transformation[2]);                                /* (2,0) */
  x[5] = atan2((double)transformation[1], (double)transformation[0]);   /* (1,0), (0,0) */
  bfgs_t B;
  bfgs_init(&B, P, x);
  int inner = 0, result = BFGS_RUNNING;
  do {
    inner++;
    result = bfgs_step(&B, x);
    if (result) break;
    result = (nrm6(B.gradient) < GICP_GRADIENT_TOL) ? BFGS_SUCCESS : BFGS_RUNNING;
  } while (result == BFGS_RUNNING && inner < GICP_MAX_INNER);
  if (result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || inner == GICP_MAX_INNER) {
    mat4f_identity(transformation);
    apply_state(transformation, x);
    return 0;
  }
  return -2;
}

/* ------------------------------------------------------------------------------------------ */
/* computeTransformation                                                                       */
/* ------------------------------------------------------------------------------------------ */
int orc_gicp_align(const float* src, size_t n_s, const float* tgt, size_t n_t, const orc_params* P, const float* guess_in,
                   float* out_xyzw, int want_fitness, orc_result* res, orc_iter_trace* trace) {
  memset(res, 0, sizeof(*res));
  mat4f_identity(res->T);
  res->fitness = NAN;
  res->convergence_state = ORC_NOT_CONVERGED;
  float guess[16];
  if (guess_in) memcpy(guess, guess_in, sizeof(guess));
  else mat4f_identity(guess);
  if (n_t == 0 || !tgt) { /* setInputTarget refuses an empty cloud */
    if (out_xyzw && n_s) orc_transform_cloud(src, n_s, res->T, out_xyzw);
    return 0;
  }
  if (n_s < GICP_K || n_t < GICP_K) { /* computeCovariances refuses clouds smaller than k_correspondences_ */
    if (out_xyzw && n_s) orc_transform_cloud(src, n_s, res->T, out_xyzw);
    return 0;
  }
  double* Ct = (double*)malloc(n_t * 9 * sizeof(double));
  double* Cs = (double*)malloc(n_s * 9 * sizeof(double));
  double* maha = (double*)malloc(n_s * 9 * sizeof(double));
  int32_t* si = (int32_t*)malloc(n_s * sizeof(int32_t));
  int32_t* ti = (int32_t*)malloc(n_s * sizeof(int32_t));
  const int seq = P->gicp_sums == ORC_GICP_SUMS_SEQUENTIAL ? 1 : P->gicp_sums == ORC_GICP_SUMS_SEQUENTIAL_REVERSED ? 2 : 0;
  orc_gicp_covariances_ex(tgt, n_t, P->arith, seq, Ct);
  orc_gicp_covariances_ex(src, n_s, P->arith, seq, Cs);
  void* tree = orc_kd_build(tgt, n_t, P->arith);
  for (size_t i = 0; i < n_s; ++i) {
    double* M = maha + 9 * i;
    for (int e = 0; e < 9; ++e) M[e] = (e % 4 == 0) ? 1.0 : 0.0;
  }
  float transformation[16], previous[16];
  mat4f_identity(transformation);
  mat4f_identity(previous);
  const double dist_threshold = P->max_correspondence_distance * P->max_correspondence_distance;
  int nr = 0, converged = 0;
  unsigned cnt = 0;
  double mse = 0.0;
  gicp_problem prob;
  prob.src = src; prob.tgt = tgt; prob.si = si; prob.ti = ti; prob.maha = maha;
  memcpy(prob.base, guess, sizeof(guess));
  prob.sequential = seq;
  prob.smooth = P->gicp_sums == ORC_GICP_SUMS_SMOOTH;
  tl_trig_cr = seq == 0; /* EXACT: correctly rounded sines / cosines; PCL's evaluation: this platform's libm */
  while (!converged) {
    float TG[16];
    mat4f_mul(transformation, guess, TG); /* query = transformation * (guess * p): applied as one float matrix */
    double R[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += (double)transformation[k * 4 + r] * (double)guess[c * 4 + k];
        R[3 * r + c] = s;
      }
    cnt = 0;
    double d2sum = 0.0;
    for (size_t i = 0; i < n_s; ++i) {
      float q[4];
      xform_point(TG, src + 4 * i, q);
      int32_t j;
      float d2;
      orc_kd_nearest(tree, q, &j, &d2);
      if (j >= 0 && (double)d2 < dist_threshold) { /* GICP: strict < */
        const double* C1 = Cs + 9 * i;
        const double* C2 = Ct + 9 * (size_t)j;
        double RC[9], tmp[9];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) RC[3 * r + c] = R[3 * r] * C1[c] + R[3 * r + 1] * C1[3 + c] + R[3 * r + 2] * C1[6 + c];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c)
            tmp[3 * r + c] = RC[3 * r] * R[3 * c] + RC[3 * r + 1] * R[3 * c + 1] + RC[3 * r + 2] * R[3 * c + 2] + C2[3 * r + c];
        inv3(tmp, maha + 9 * i);
        if (!seq) {
          /* R C1 R^T + C2 is symmetric only up to rounding, and so is its adjugate inverse; PCL keeps all nine entries.
           * The upper triangle is mirrored here: the difference is in the last bit of three entries, and an
           * implementation that stores six numbers per correspondence (the GPU path: 48 instead of 72 B of the 88 B an
           * evaluation reads per correspondence) can then be compared bit for bit. */
          double* Mi = maha + 9 * i;
          Mi[3] = Mi[1];
          Mi[6] = Mi[2];
          Mi[7] = Mi[5];
        }
        si[cnt] = (int32_t)i;
        ti[cnt] = j;
        d2sum += (double)d2;
        cnt++;
      }
    }
    prob.m = (int)cnt;
    mse = cnt ? d2sum / cnt : 0.0;
    memcpy(previous, transformation, sizeof(previous));
    const int rc = estimate_bfgs(&prob, transformation);
    if (rc != 0) { /* NotEnoughPoints / SolverDidntConverge: PCL catches the exception and breaks, converged_ stays false */
      res->convergence_state = rc == -1 ? ORC_NO_CORRESPONDENCES : ORC_NOT_CONVERGED;
      break;
    }
    double delta = 0.0;
    for (int k = 0; k < 4; ++k)
      for (int l = 0; l < 4; ++l) {
        const double ratio = (k < 3 && l < 3) ? 1.0 / GICP_ROTATION_EPSILON : 1.0 / P->transformation_epsilon;
        const double c_delta = ratio * fabs((double)previous[l * 4 + k] - (double)transformation[l * 4 + k]);
        if (c_delta > delta) delta = c_delta;
      }
    if (trace) {
      orc_iter_trace* tr = &trace[nr];
      memset(tr, 0, sizeof(*tr));
      for (int e = 0; e < 16; ++e) tr->Tk[e] = transformation[e];
      for (int e = 0; e < 16; ++e) tr->final[e] = transformation[e];
      tr->n_corr = cnt;
      tr->mse = mse;
    }
    nr++;
    if (nr >= P->max_iterations || (delta < 1 && !P->force_iterations)) {
      converged = 1;
      res->convergence_state = nr >= P->max_iterations ? ORC_ITERATIONS : ORC_TRANSFORM;
      memcpy(previous, transformation, sizeof(previous));
    }
  }
  /* final = previous.R * guess.R ; t = previous.t + guess.t  (PCL's own composition, not a matrix product) */
  float final[16];
  mat4f_identity(final);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += previous[k * 4 + r] * guess[c * 4 + k];
      final[c * 4 + r] = s;
    }
    final[12 + r] = previous[12 + r] + guess[12 + r];
  }
  memcpy(res->T, final, sizeof(final));
  res->converged = converged;
  res->iterations = nr;
  res->n_correspondences = cnt;
  res->mse_last = mse;
  if (out_xyzw) orc_transform_cloud(src, n_s, final, out_xyzw);
  if (want_fitness) res->fitness = orc_fitness(src, n_s, tgt, n_t, final, DBL_MAX, ORC_NN_KDTREE, P->arith);
  orc_kd_free(tree);
  free(Ct); free(Cs); free(maha); free(si); free(ti);
  return 0;
}
