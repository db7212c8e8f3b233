/*
 * icp_oracle.c -- CPU ORACLE (test infrastructure; see icp_oracle.h header comment).
 * PARITY UNPINNED: restates upstream PCL 1.8.x semantics; the
 * reference (/root/reference) holds only the call sites and parameters:
 *   src/icpslam/icp_odometer.cpp:188-201, include/icpslam/icp_odometer.h:62-65,
 *   src/icpslam/octree_mapper.cpp:104-117, include/icpslam/octree_mapper.h:53-56.
 */
#include "icp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* hot loops are cloned for FMA hardware (inline vfmadd) with a portable default (libm fmaf):
 * same results either way, fmaf is exactly rounded. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORC_HOT __attribute__((target_clones("fma", "default")))
#else
#define ORC_HOT
#endif

/* ------------------------------------------------------------------------------------------ */
/* small linear algebra (double, column-major 4x4 / row-major 3x3)                            */
/* ------------------------------------------------------------------------------------------ */

static void mat4_identity(double M[16]) {
  memset(M, 0, 16 * sizeof(double));
  M[0] = M[5] = M[10] = M[15] = 1.0;
}

/* C = A*B, column-major */
static void mat4_mul(const double A[16], const double B[16], double C[16]) {
  double R[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
      R[c * 4 + r] = s;
    }
  memcpy(C, R, sizeof(R));
}

static void mat4f_mul(const float A[16], const float B[16], float C[16]) {
  float R[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      float s = 0.0f;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + r] * B[c * 4 + k];
      R[c * 4 + r] = s;
    }
  memcpy(C, R, sizeof(R));
}

static double det3(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}

/* One-sided (Hestenes) Jacobi SVD of a 3x3 row-major matrix: A = U diag(s) V^T. */
void orc_svd3(const double A[9], double U[9], double s[3], double V[9]) {
  double W[9]; /* working copy, columns get orthogonalised: W = A*V */
  memcpy(W, A, sizeof(W));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;

  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += W[r * 3 + p] * W[r * 3 + p];
          beta += W[r * 3 + q] * W[r * 3 + q];
          gamma += W[r * 3 + p] * W[r * 3 + q];
        }
        if (gamma == 0.0) continue;
        double lim = sqrt(alpha * beta);
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * lim) continue;
        off = fmax(off, fabs(gamma) / (lim > 0 ? lim : 1.0));
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < 3; ++r) {
          double wp = W[r * 3 + p], wq = W[r * 3 + q];
          W[r * 3 + p] = c * wp - sn * wq;
          W[r * 3 + q] = sn * wp + c * wq;
          double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - sn * vq;
          V[r * 3 + q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-16) break;
  }
  /* singular values = column norms; sort descending */
  double nrm[3];
  int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j)
    nrm[j] = sqrt(W[0 * 3 + j] * W[0 * 3 + j] + W[1 * 3 + j] * W[1 * 3 + j] + W[2 * 3 + j] * W[2 * 3 + j]);
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (nrm[ord[b]] > nrm[ord[a]]) {
        int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
      }
  double Vs[9], Us[9];
  for (int j = 0; j < 3; ++j) {
    int o = ord[j];
    s[j] = nrm[o];
    for (int r = 0; r < 3; ++r) {
      Vs[r * 3 + j] = V[r * 3 + o];
      Us[r * 3 + j] = (nrm[o] > 0) ? W[r * 3 + o] / nrm[o] : 0.0;
    }
  }
  /* complete U for (near-)zero singular values so that U is orthonormal */
  double tiny = 1e-13 * (s[0] > 0 ? s[0] : 1.0);
  if (s[0] <= tiny) {
    for (int i = 0; i < 9; ++i) Us[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    if (s[1] <= tiny) {
      /* pick any unit vector orthogonal to u0 */
      double u0[3] = {Us[0], Us[3], Us[6]};
      int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0 : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
      double e[3] = {0, 0, 0};
      e[k] = 1.0;
      double d = u0[k];
      double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
      double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      Us[1] = v[0] / n;
      Us[4] = v[1] / n;
      Us[7] = v[2] / n;
    }
    if (s[2] <= tiny) {
      double a[3] = {Us[0], Us[3], Us[6]}, b[3] = {Us[1], Us[4], Us[7]};
      Us[2] = a[1] * b[2] - a[2] * b[1];
      Us[5] = a[2] * b[0] - a[0] * b[2];
      Us[8] = a[0] * b[1] - a[1] * b[0];
    }
  }
  memcpy(U, Us, sizeof(Us));
  memcpy(V, Vs, sizeof(Vs));
}

/* ------------------------------------------------------------------------------------------ */
/* shared arithmetic contract                                                                  */
/* ------------------------------------------------------------------------------------------ */

static inline void xform_point(const float T[16], const float* s, float* p) {
  p[0] = fmaf(T[8], s[2], fmaf(T[4], s[1], fmaf(T[0], s[0], T[12])));
  p[1] = fmaf(T[9], s[2], fmaf(T[5], s[1], fmaf(T[1], s[0], T[13])));
  p[2] = fmaf(T[10], s[2], fmaf(T[6], s[1], fmaf(T[2], s[0], T[14])));
}

static inline float dist2_fma(const float* q, const float* p) {
  float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* FLANN L2_Simple: result += diff*diff, no contraction (volatile blocks gcc's -ffp-contract) */
static inline float dist2_flann(const float* q, const float* p) {
  volatile float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
  volatile float a = dx * dx;
  volatile float b = dy * dy;
  volatile float c = dz * dz;
  volatile float r = a + b;
  r = r + c;
  return r;
}

ORC_HOT void orc_transform_cloud(const float* in, size_t n, const float T[16], float* out) {
  for (size_t i = 0; i < n; ++i) {
    float p[3];
    xform_point(T, in + 4 * i, p);
    out[4 * i + 0] = p[0];
    out[4 * i + 1] = p[1];
    out[4 * i + 2] = p[2];
    out[4 * i + 3] = 1.0f;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* exact kd-tree (stands in for FLANN KDTreeSingleIndex(leaf 15), exact search)                */
/* ------------------------------------------------------------------------------------------ */

#define KD_LEAF 15

typedef struct {
  int32_t left, right; /* children (node ids); -1 for leaf */
  int32_t lo, hi;      /* range in perm (leaf) */
  int32_t axis;
  float divlow, divhigh; /* max of left subtree / min of right subtree on axis */
} kd_node;

typedef struct {
  const float* pts; /* xyzw */
  size_t n;
  int32_t* perm;
  float* packed; /* xyz of perm order, 4 floats each (x,y,z,idx-as-bits) for locality */
  kd_node* nodes;
  int32_t n_nodes, cap_nodes;
  float bbox_lo[3], bbox_hi[3];
  int arith;
} kd_tree;

static int32_t kd_new_node(kd_tree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kd_node*)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(kd_node));
  }
  return t->n_nodes++;
}

static inline float kd_coord(const kd_tree* t, int32_t pi, int axis) { return t->pts[4 * (size_t)pi + axis]; }

/* quickselect: arrange perm[lo..hi) so that element at k is in sorted position by (coord, index) */
static void kd_select(kd_tree* t, int32_t lo, int32_t hi, int32_t k, int axis) {
  int32_t* p = t->perm;
  while (hi - lo > 1) {
    int32_t mid = lo + (hi - lo) / 2;
    /* median of three pivot */
    int32_t a = p[lo], b = p[mid], c = p[hi - 1];
    float fa = kd_coord(t, a, axis), fb = kd_coord(t, b, axis), fc = kd_coord(t, c, axis);
    int32_t piv;
    if ((fa <= fb && fb <= fc) || (fc <= fb && fb <= fa))
      piv = b;
    else if ((fb <= fa && fa <= fc) || (fc <= fa && fa <= fb))
      piv = a;
    else
      piv = c;
    float fp = kd_coord(t, piv, axis);
    int32_t i = lo, j = hi - 1;
    while (i <= j) {
      while (kd_coord(t, p[i], axis) < fp || (kd_coord(t, p[i], axis) == fp && p[i] < piv)) ++i;
      while (kd_coord(t, p[j], axis) > fp || (kd_coord(t, p[j], axis) == fp && p[j] > piv)) --j;
      if (i <= j) {
        int32_t tmp = p[i];
        p[i] = p[j];
        p[j] = tmp;
        ++i;
        --j;
      }
    }
    if (k <= j)
      hi = j + 1;
    else if (k >= i)
      lo = i;
    else
      return;
  }
}

static int32_t kd_build_rec(kd_tree* t, int32_t lo, int32_t hi) {
  int32_t id = kd_new_node(t);
  if (hi - lo <= KD_LEAF) {
    kd_node* nd = &t->nodes[id];
    nd->left = nd->right = -1;
    nd->lo = lo;
    nd->hi = hi;
    nd->axis = 0;
    nd->divlow = nd->divhigh = 0;
    return id;
  }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int32_t i = lo; i < hi; ++i)
    for (int a = 0; a < 3; ++a) {
      float v = kd_coord(t, t->perm[i], a);
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int axis = 0;
  float ext = mx[0] - mn[0];
  for (int a = 1; a < 3; ++a)
    if (mx[a] - mn[a] > ext) {
      ext = mx[a] - mn[a];
      axis = a;
    }
  int32_t mid = lo + (hi - lo) / 2;
  kd_select(t, lo, hi, mid, axis);
  float dl = -FLT_MAX, dh = FLT_MAX;
  for (int32_t i = lo; i < mid; ++i) {
    float v = kd_coord(t, t->perm[i], axis);
    if (v > dl) dl = v;
  }
  for (int32_t i = mid; i < hi; ++i) {
    float v = kd_coord(t, t->perm[i], axis);
    if (v < dh) dh = v;
  }
  int32_t l = kd_build_rec(t, lo, mid);
  int32_t r = kd_build_rec(t, mid, hi);
  kd_node* nd = &t->nodes[id]; /* re-fetch: realloc may have moved */
  nd->left = l;
  nd->right = r;
  nd->lo = lo;
  nd->hi = hi;
  nd->axis = axis;
  nd->divlow = dl;
  nd->divhigh = dh;
  return id;
}

static kd_tree* kd_build(const float* pts, size_t n, int arith) {
  kd_tree* t = (kd_tree*)calloc(1, sizeof(kd_tree));
  t->pts = pts;
  t->n = n;
  t->arith = arith;
  t->perm = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
  /* non-finite points never enter the tree but keep their indices: pcl::KdTreeFLANN::convertCloudToArray skips points
   * for which the point representation is not valid and maps tree indices back to cloud indices */
  size_t m = 0;
  for (size_t i = 0; i < n; ++i)
    if (isfinite(pts[4 * i]) && isfinite(pts[4 * i + 1]) && isfinite(pts[4 * i + 2])) t->perm[m++] = (int32_t)i;
  n = m;
  t->n = n;
  for (int a = 0; a < 3; ++a) {
    t->bbox_lo[a] = FLT_MAX;
    t->bbox_hi[a] = -FLT_MAX;
  }
  for (size_t k = 0; k < n; ++k)
    for (int a = 0; a < 3; ++a) {
      float v = pts[4 * (size_t)t->perm[k] + a];
      if (v < t->bbox_lo[a]) t->bbox_lo[a] = v;
      if (v > t->bbox_hi[a]) t->bbox_hi[a] = v;
    }
  if (n) kd_build_rec(t, 0, (int32_t)n);
  t->packed = (float*)malloc((n ? n : 1) * 4 * sizeof(float));
  for (size_t i = 0; i < n; ++i) {
    int32_t pi = t->perm[i];
    t->packed[4 * i + 0] = pts[4 * (size_t)pi + 0];
    t->packed[4 * i + 1] = pts[4 * (size_t)pi + 1];
    t->packed[4 * i + 2] = pts[4 * (size_t)pi + 2];
    memcpy(&t->packed[4 * i + 3], &pi, 4);
  }
  return t;
}

static void kd_free(kd_tree* t) {
  if (!t) return;
  free(t->perm);
  free(t->packed);
  free(t->nodes);
  free(t);
}

typedef struct {
  const kd_tree* t;
  const float* q;
  float best;
  int32_t best_idx;
} kd_query;

ORC_HOT static void kd_search_rec(kd_query* Q, int32_t id, double mindist, double offs[3]) {
  const kd_tree* t = Q->t;
  const kd_node* nd = &t->nodes[id];
  if (nd->left < 0) {
    for (int32_t i = nd->lo; i < nd->hi; ++i) {
      const float* p = &t->packed[4 * (size_t)i];
      float d = t->arith == ORC_ARITH_FLANN ? dist2_flann(p, Q->q) : dist2_fma(p, Q->q);
      int32_t pi;
      memcpy(&pi, &p[3], 4);
      if (d < Q->best || (d == Q->best && pi < Q->best_idx)) {
        Q->best = d;
        Q->best_idx = pi;
      }
    }
    return;
  }
  int axis = nd->axis;
  double val = Q->q[axis];
  double d_lo = val - (double)nd->divlow, d_hi = val - (double)nd->divhigh;
  int32_t near, far;
  double cut;
  if (d_lo + d_hi < 0) {
    near = nd->left;
    far = nd->right;
    cut = d_hi;
  } else {
    near = nd->right;
    far = nd->left;
    cut = d_lo;
  }
  kd_search_rec(Q, near, mindist, offs);
  double old = offs[axis];
  double nd2 = mindist - old * old + cut * cut;
  /* conservative prune: float d2 may sit up to ~3 ulp below the exact value; keep ties */
  if (nd2 * (1.0 - 1e-6) <= (double)Q->best) {
    offs[axis] = cut;
    kd_search_rec(Q, far, nd2, offs);
    offs[axis] = old;
  }
}

static void kd_nearest(const kd_tree* t, const float* q, int32_t* idx, float* d2) {
  kd_query Q = {t, q, INFINITY, INT32_MAX};
  if (t->n == 0) {
    *idx = -1;
    *d2 = INFINITY;
    return;
  }
  double offs[3] = {0, 0, 0}, mind = 0.0;
  for (int a = 0; a < 3; ++a) {
    if (q[a] < t->bbox_lo[a]) offs[a] = (double)q[a] - (double)t->bbox_lo[a];
    if (q[a] > t->bbox_hi[a]) offs[a] = (double)q[a] - (double)t->bbox_hi[a];
    mind += offs[a] * offs[a];
  }
  kd_search_rec(&Q, 0, mind, offs);
  *idx = Q.best_idx == INT32_MAX ? -1 : Q.best_idx; /* non-finite query: no neighbour (like the brute-force scan) */
  *d2 = Q.best;
}

/* ---- k nearest neighbours (GICP covariances): the k smallest (d2, index) keys, ascending ---- */
typedef struct {
  const kd_tree* t;
  const float* q;
  int k, n;
  float* d2;     /* sorted ascending by (d2, idx) */
  int32_t* idx;
} kd_knn_query;

static inline int knn_less(float da, int32_t ia, float db, int32_t ib) { return da < db || (da == db && ia < ib); }

static inline void knn_insert(kd_knn_query* Q, float d, int32_t pi) {
  if (Q->n == Q->k && !knn_less(d, pi, Q->d2[Q->k - 1], Q->idx[Q->k - 1])) return;
  int pos = Q->n < Q->k ? Q->n : Q->k - 1;
  while (pos > 0 && knn_less(d, pi, Q->d2[pos - 1], Q->idx[pos - 1])) {
    Q->d2[pos] = Q->d2[pos - 1];
    Q->idx[pos] = Q->idx[pos - 1];
    --pos;
  }
  Q->d2[pos] = d;
  Q->idx[pos] = pi;
  if (Q->n < Q->k) Q->n++;
}

ORC_HOT static void kd_knn_rec(kd_knn_query* Q, int32_t id, double mindist, double offs[3]) {
  const kd_tree* t = Q->t;
  const kd_node* nd = &t->nodes[id];
  if (nd->left < 0) {
    for (int32_t i = nd->lo; i < nd->hi; ++i) {
      const float* p = &t->packed[4 * (size_t)i];
      float d = t->arith == ORC_ARITH_FLANN ? dist2_flann(p, Q->q) : dist2_fma(p, Q->q);
      int32_t pi;
      memcpy(&pi, &p[3], 4);
      knn_insert(Q, d, pi);
    }
    return;
  }
  int axis = nd->axis;
  double val = Q->q[axis];
  double d_lo = val - (double)nd->divlow, d_hi = val - (double)nd->divhigh;
  int32_t near, far;
  double cut;
  if (d_lo + d_hi < 0) {
    near = nd->left;
    far = nd->right;
    cut = d_hi;
  } else {
    near = nd->right;
    far = nd->left;
    cut = d_lo;
  }
  kd_knn_rec(Q, near, mindist, offs);
  double old = offs[axis];
  double nd2 = mindist - old * old + cut * cut;
  if (Q->n < Q->k || nd2 * (1.0 - 1e-6) <= (double)Q->d2[Q->k - 1]) {
    offs[axis] = cut;
    kd_knn_rec(Q, far, nd2, offs);
    offs[axis] = old;
  }
}

/* internal (oracle_internal.h): opaque kd-tree handle for gicp_oracle.c */
void* orc_kd_build(const float* pts, size_t n, int arith) { return kd_build(pts, n, arith); }
void orc_kd_free(void* t) { kd_free((kd_tree*)t); }
void orc_kd_nearest(const void* t, const float* q, int32_t* idx, float* d2) { kd_nearest((const kd_tree*)t, q, idx, d2); }
int orc_kd_knn(const void* tv, const float* q, int k, int32_t* idx, float* d2) {
  const kd_tree* t = (const kd_tree*)tv;
  kd_knn_query Q = {t, q, k, 0, d2, idx};
  if (t->n == 0 || k <= 0) return 0;
  double offs[3] = {0, 0, 0}, mind = 0.0;
  for (int a = 0; a < 3; ++a) {
    if (q[a] < t->bbox_lo[a]) offs[a] = (double)q[a] - (double)t->bbox_lo[a];
    if (q[a] > t->bbox_hi[a]) offs[a] = (double)q[a] - (double)t->bbox_hi[a];
    mind += offs[a] * offs[a];
  }
  kd_knn_rec(&Q, 0, mind, offs);
  return Q.n;
}

ORC_HOT static void brute_nearest(const float* tgt, size_t n_t, const float* q, int arith, int32_t* idx, float* d2) {
  float best = INFINITY;
  int32_t bi = -1;
  for (size_t j = 0; j < n_t; ++j) {
    float d = arith == ORC_ARITH_FLANN ? dist2_flann(tgt + 4 * j, q) : dist2_fma(tgt + 4 * j, q);
    if (d < best) { /* strict: first (lowest) index wins ties */
      best = d;
      bi = (int32_t)j;
    }
  }
  *idx = bi;
  *d2 = best;
}

/* NN of every (already transformed) X[i] in tgt */
static void nn_all(const float* X, size_t n_s, const float* tgt, size_t n_t, const kd_tree* tree, int arith,
                   int32_t* idx, float* d2) {
  for (size_t i = 0; i < n_s; ++i) {
    if (tree)
      kd_nearest(tree, X + 4 * i, &idx[i], &d2[i]);
    else
      brute_nearest(tgt, n_t, X + 4 * i, arith, &idx[i], &d2[i]);
  }
}

int orc_nn(const float* src, size_t n_s, const float* tgt, size_t n_t, const float T[16], int nn_mode, int arith,
           int32_t* idx, float* d2) {
  float* X = (float*)malloc((n_s ? n_s : 1) * 4 * sizeof(float));
  orc_transform_cloud(src, n_s, T, X);
  kd_tree* tree = (nn_mode == ORC_NN_KDTREE) ? kd_build(tgt, n_t, arith) : NULL;
  nn_all(X, n_s, tgt, n_t, tree, arith, idx, d2);
  kd_free(tree);
  free(X);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* a3 + a4: rejection and reduction                                                            */
/* ------------------------------------------------------------------------------------------ */

static void reduce_X(const float* X, size_t n_s, const float* tgt, const int32_t* idx, const float* d2,
                     double max_corr_dist, double sums[17]) {
  double r2 = max_corr_dist * max_corr_dist;
  for (int k = 0; k < 17; ++k) sums[k] = 0.0;
  for (size_t i = 0; i < n_s; ++i) {
    if (idx[i] < 0) continue;
    if ((double)d2[i] > r2) continue; /* PCL P2P: "if (distance[0] > max_dist_sqr) continue;" */
    const float* p = X + 4 * i;
    const float* q = tgt + 4 * (size_t)idx[i];
    sums[0] += 1.0;
    for (int a = 0; a < 3; ++a) {
      sums[1 + a] += (double)p[a];
      sums[4 + a] += (double)q[a];
      for (int b = 0; b < 3; ++b) sums[7 + 3 * a + b] += (double)q[a] * (double)p[b];
    }
    sums[16] += (double)d2[i];
  }
}

int orc_reduce(const float* src, size_t n_s, const float* tgt, const float T[16], const int32_t* idx,
               const float* d2, double max_corr_dist, double sums[17]) {
  float* X = (float*)malloc((n_s ? n_s : 1) * 4 * sizeof(float));
  orc_transform_cloud(src, n_s, T, X);
  reduce_X(X, n_s, tgt, idx, d2, max_corr_dist, sums);
  free(X);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* a5: Umeyama without scaling (Eigen::umeyama(src, dst, false))                               */
/* ------------------------------------------------------------------------------------------ */

static int umeyama_from_cov(const double mu_p[3], const double mu_q[3], const double Sigma[9], double Tk[16]) {
  double U[9], s[3], V[9];
  orc_svd3(Sigma, U, s, V);
  double S[3] = {1.0, 1.0, 1.0};
  if (det3(U) * det3(V) < 0.0) S[2] = -1.0;
  double R[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += U[r * 3 + k] * S[k] * V[c * 3 + k];
      R[r * 3 + c] = acc;
    }
  mat4_identity(Tk);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Tk[c * 4 + r] = R[r * 3 + c];
    Tk[12 + r] = mu_q[r] - (R[r * 3 + 0] * mu_p[0] + R[r * 3 + 1] * mu_p[1] + R[r * 3 + 2] * mu_p[2]);
  }
  for (int i = 0; i < 16; ++i)
    if (!isfinite(Tk[i])) return -1;
  return 0;
}

int orc_umeyama(const double sums[17], double Tk[16]) {
  double n = sums[0];
  if (n < 1.0) {
    mat4_identity(Tk);
    return -1;
  }
  double mu_p[3], mu_q[3], Sigma[9];
  for (int a = 0; a < 3; ++a) {
    mu_p[a] = sums[1 + a] / n;
    mu_q[a] = sums[4 + a] / n;
  }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) Sigma[3 * a + b] = sums[7 + 3 * a + b] / n - mu_q[a] * mu_p[b];
  return umeyama_from_cov(mu_p, mu_q, Sigma, Tk);
}

/* PCL-float flavour: two-pass (means, then demeaned covariance), float accumulators. */
static int umeyama_f32(const float* X, size_t n_s, const float* tgt, const int32_t* idx, const float* d2, double r2,
                       double Tk[16]) {
  float sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
  size_t n = 0;
  for (size_t i = 0; i < n_s; ++i) {
    if (idx[i] < 0 || (double)d2[i] > r2) continue;
    for (int a = 0; a < 3; ++a) {
      sp[a] += X[4 * i + a];
      sq[a] += tgt[4 * (size_t)idx[i] + a];
    }
    ++n;
  }
  if (!n) return -1;
  float inv = 1.0f / (float)n;
  float mp[3] = {sp[0] * inv, sp[1] * inv, sp[2] * inv}, mq[3] = {sq[0] * inv, sq[1] * inv, sq[2] * inv};
  float Sg[9] = {0};
  for (size_t i = 0; i < n_s; ++i) {
    if (idx[i] < 0 || (double)d2[i] > r2) continue;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Sg[3 * a + b] += (tgt[4 * (size_t)idx[i] + a] - mq[a]) * (X[4 * i + b] - mp[b]);
  }
  double mu_p[3], mu_q[3], Sigma[9];
  for (int a = 0; a < 3; ++a) {
    mu_p[a] = mp[a];
    mu_q[a] = mq[a];
  }
  for (int k = 0; k < 9; ++k) Sigma[k] = (double)(Sg[k] * inv);
  int rc = umeyama_from_cov(mu_p, mu_q, Sigma, Tk);
  for (int k = 0; k < 16; ++k) Tk[k] = (double)(float)Tk[k];
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* a9: fitness                                                                                 */
/* ------------------------------------------------------------------------------------------ */

static double fitness_X(const float* X, size_t n_s, const float* tgt, size_t n_t, const kd_tree* tree, int arith,
                        double max_range) {
  double acc = 0.0;
  size_t nr = 0;
  for (size_t i = 0; i < n_s; ++i) {
    int32_t j;
    float d;
    if (tree)
      kd_nearest(tree, X + 4 * i, &j, &d);
    else
      brute_nearest(tgt, n_t, X + 4 * i, arith, &j, &d);
    if (j >= 0 && (double)d <= max_range) {
      acc += (double)d;
      ++nr;
    }
  }
  return nr ? acc / (double)nr : DBL_MAX;
}

double orc_fitness(const float* src, size_t n_s, const float* tgt, size_t n_t, const float T[16], double max_range,
                   int nn_mode, int arith) {
  float* X = (float*)malloc((n_s ? n_s : 1) * 4 * sizeof(float));
  orc_transform_cloud(src, n_s, T, X);
  kd_tree* tree = (nn_mode == ORC_NN_KDTREE) ? kd_build(tgt, n_t, arith) : NULL;
  double f = fitness_X(X, n_s, tgt, n_t, tree, arith, max_range);
  kd_free(tree);
  free(X);
  return f;
}

/* ------------------------------------------------------------------------------------------ */
/* a1..a10: Registration::align / IterativeClosestPoint::computeTransformation                 */
/* ------------------------------------------------------------------------------------------ */

void orc_default_params(orc_params* p) {
  p->method = ORC_P2P_SVD;
  p->max_iterations = 10;
  p->transformation_epsilon = 1e-6;
  p->max_correspondence_distance = 1.0;
  p->euclidean_fitness_epsilon = -DBL_MAX;
  p->min_correspondences = 3;
  p->force_iterations = 0;
  p->nn_mode = ORC_NN_KDTREE;
  p->precision = ORC_PREC_F64;
  p->arith = ORC_ARITH_FMA;
  p->gicp_sums = ORC_GICP_SUMS_EXACT;
}

int orc_gicp_align(const float* src, size_t n_s, const float* tgt, size_t n_t, const orc_params* P,
                   const float* guess, float* out_xyzw, int want_fitness, orc_result* res, orc_iter_trace* trace);

int orc_icp_align(const float* src, size_t n_s, const float* tgt, size_t n_t, const orc_params* P, const float* guess,
                  float* out_xyzw, int want_fitness, orc_result* res, orc_iter_trace* trace) {
  if (!P || !res) return -1;
  if (P->method == ORC_GICP) return orc_gicp_align(src, n_s, tgt, n_t, P, guess, out_xyzw, want_fitness, res, trace);

  memset(res, 0, sizeof(*res));
  double final[16];
  mat4_identity(final);
  for (int i = 0; i < 16; ++i) res->T[i] = (float)final[i];
  res->fitness = NAN;
  res->convergence_state = ORC_NOT_CONVERGED;

  /* Registration::setInputTarget rejects an empty target -> initCompute fails -> align returns early */
  if (n_t == 0 || !tgt) {
    if (out_xyzw && n_s) orc_transform_cloud(src, n_s, res->T, out_xyzw);
    return 0;
  }

  float finalf[16];
  if (guess) {
    for (int i = 0; i < 16; ++i) {
      final[i] = (double)guess[i];
      finalf[i] = guess[i];
    }
  } else
    for (int i = 0; i < 16; ++i) finalf[i] = (float)final[i];

  size_t nalloc = n_s ? n_s : 1;
  float* X = (float*)malloc(nalloc * 4 * sizeof(float));
  int32_t* idx = (int32_t*)malloc(nalloc * sizeof(int32_t));
  float* d2 = (float*)malloc(nalloc * sizeof(float));
  kd_tree* tree = (P->nn_mode == ORC_NN_KDTREE) ? kd_build(tgt, n_t, P->arith) : NULL;

  orc_transform_cloud(src, n_s, finalf, X);

  const double r2 = P->max_correspondence_distance * P->max_correspondence_distance;
  const double rotation_thr = 1.0 - P->transformation_epsilon;
  const double translation_thr = P->transformation_epsilon;
  const double mse_abs_thr = 1e-12;
  const double mse_rel_thr = P->euclidean_fitness_epsilon;
  double mse_prev = DBL_MAX;
  int nr_iter = 0, converged = 0, state = ORC_NOT_CONVERGED;
  unsigned n_corr = 0;
  double mse = 0.0;

  do {
    nn_all(X, n_s, tgt, n_t, tree, P->arith, idx, d2);
    double sums[17];
    reduce_X(X, n_s, tgt, idx, d2, P->max_correspondence_distance, sums);
    n_corr = (unsigned)sums[0];
    if ((int)n_corr < P->min_correspondences) {
      state = ORC_NO_CORRESPONDENCES;
      converged = 0;
      break;
    }
    double Tk[16];
    int rc = (P->precision == ORC_PREC_PCL_F32) ? umeyama_f32(X, n_s, tgt, idx, d2, r2, Tk) : orc_umeyama(sums, Tk);
    if (rc != 0) {
      state = ORC_NO_CORRESPONDENCES;
      converged = 0;
      break;
    }
    if (P->precision == ORC_PREC_PCL_F32) {
      /* PCL: transformCloud(X, X, T_k) in place; final = T_k * final in Matrix4f */
      float Tkf[16], tmp[16];
      for (int i = 0; i < 16; ++i) Tkf[i] = (float)Tk[i];
      orc_transform_cloud(X, n_s, Tkf, X);
      mat4f_mul(Tkf, finalf, tmp);
      memcpy(finalf, tmp, sizeof(tmp));
      for (int i = 0; i < 16; ++i) final[i] = (double)finalf[i];
    } else {
      mat4_mul(Tk, final, final);
      for (int i = 0; i < 16; ++i) finalf[i] = (float)final[i];
      orc_transform_cloud(src, n_s, finalf, X);
    }
    mse = sums[16] / sums[0];
    if (trace) {
      orc_iter_trace* tr = &trace[nr_iter];
      memcpy(tr->Tk, Tk, sizeof(tr->Tk));
      memcpy(tr->final, final, sizeof(tr->final));
      memcpy(tr->sums, sums, sizeof(tr->sums));
      tr->n_corr = n_corr;
      tr->mse = mse;
    }
    ++nr_iter;

    /* DefaultConvergenceCriteria::hasConverged (max_iterations_similar_transforms_ = 0) */
    converged = 0;
    if (nr_iter >= P->max_iterations) {
      converged = 1;
      state = ORC_ITERATIONS;
    } else if (!P->force_iterations) {
      double cos_angle = 0.5 * (Tk[0] + Tk[5] + Tk[10] - 1.0);
      double tsq = Tk[12] * Tk[12] + Tk[13] * Tk[13] + Tk[14] * Tk[14];
      if (cos_angle >= rotation_thr && tsq <= translation_thr) {
        converged = 1;
        state = ORC_TRANSFORM;
      } else if (fabs(mse - mse_prev) < mse_abs_thr) {
        converged = 1;
        state = ORC_ABS_MSE;
      } else if (fabs(mse - mse_prev) / mse_prev < mse_rel_thr) {
        converged = 1;
        state = ORC_REL_MSE;
      }
      mse_prev = mse;
    }
  } while (!converged);

  for (int i = 0; i < 16; ++i) res->T[i] = finalf[i];
  res->converged = converged;
  res->iterations = nr_iter;
  res->convergence_state = state;
  res->n_correspondences = n_corr;
  res->mse_last = mse;
  if (out_xyzw) orc_transform_cloud(src, n_s, finalf, out_xyzw);
  if (want_fitness) {
    orc_transform_cloud(src, n_s, finalf, X);
    res->fitness = fitness_X(X, n_s, tgt, n_t, tree, P->arith, DBL_MAX);
  }
  kd_free(tree);
  free(X);
  free(idx);
  free(d2);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* f2: pcl::VoxelGrid<PointXYZ>::filter (SURVEY.md Appendix A.3;                               */
/*     call site /root/reference/src/icpslam/icp_odometer.cpp:96-101)                          */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  int32_t cell;
  int32_t pt;
} cell_pt;

static int cmp_cell_pt(const void* a, const void* b) {
  const cell_pt* x = (const cell_pt*)a;
  const cell_pt* y = (const cell_pt*)b;
  if (x->cell != y->cell) return x->cell < y->cell ? -1 : 1;
  return x->pt < y->pt ? -1 : (x->pt > y->pt);
}

long orc_voxel_grid(const float* in, size_t n, float leaf, float* out) {
  if (n == 0) return 0;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      float v = in[4 * i + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  float inv = 1.0f / leaf; /* PCL: inverse_leaf_size_ = 1/leaf_size_ (float) */
  int64_t dxyz[3];
  int32_t minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; ++a) {
    dxyz[a] = (int64_t)((mx[a] - mn[a]) * inv) + 1;
    minb[a] = (int32_t)floorf(mn[a] * inv);
    maxb[a] = (int32_t)floorf(mx[a] * inv);
    divb[a] = maxb[a] - minb[a] + 1;
  }
  if (dxyz[0] * dxyz[1] * dxyz[2] > (int64_t)INT32_MAX) return -1;
  int32_t mul[3] = {1, divb[0], divb[0] * divb[1]};
  cell_pt* cp = (cell_pt*)malloc(n * sizeof(cell_pt));
  for (size_t i = 0; i < n; ++i) {
    int32_t c = 0;
    for (int a = 0; a < 3; ++a) c += ((int32_t)floorf(in[4 * i + a] * inv) - minb[a]) * mul[a];
    cp[i].cell = c;
    cp[i].pt = (int32_t)i;
  }
  qsort(cp, n, sizeof(cell_pt), cmp_cell_pt);
  long n_out = 0;
  size_t i = 0;
  while (i < n) {
    size_t j = i;
    float acc[3] = {0, 0, 0};
    while (j < n && cp[j].cell == cp[i].cell) {
      for (int a = 0; a < 3; ++a) acc[a] += in[4 * (size_t)cp[j].pt + a];
      ++j;
    }
    float cnt = (float)(j - i);
    out[4 * n_out + 0] = acc[0] / cnt;
    out[4 * n_out + 1] = acc[1] / cnt;
    out[4 * n_out + 2] = acc[2] / cnt;
    out[4 * n_out + 3] = 1.0f;
    ++n_out;
    i = j;
  }
  free(cp);
  return n_out;
}
