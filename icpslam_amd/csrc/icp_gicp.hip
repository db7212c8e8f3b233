// icp_gicp.hip -- device side of the GICP mode: what
// pcl::GeneralizedIterativeClosestPoint does per point / per correspondence, reached in the reference through
// `icp.align()` at /root/reference/src/icpslam/icp_odometer.cpp:188-198 and src/icpslam/octree_mapper.cpp:104-114.
//
//   gicp_cov_kernel   computeCovariances: 20 nearest neighbours of every point in its own cloud (the 20 smallest
//                     (d2, index) keys), covariance in double from float products, smallest singular direction u,
//                     C = I - (1 - 1e-3) u u^T  (== U diag(1, 1, 1e-3) U^T).  One wave per point over the cloud's own
//                     uniform grid: lanes 0..19 hold the running top-20, candidates stream in 64 at a time.
//   gicp_maha_kernel  per accepted correspondence (d2 < r^2, strict): M_i = (C_t[j] + R C_s[i] R^T)^-1, symmetric 6.
//   gicp_cost_kernel  one BFGS function/gradient evaluation: sum over correspondences of r^T M r, M r and
//                     (base p)(M r)^T with r = T(x) p - q -- 14 float64 terms, fixed-order two-stage reduction.
//                     HBM-bound: 16 B source + 8 B key + 16 B gathered target + 48 B M per correspondence.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <cstring>

#include "icp_env.h"
#include "icp_gicp_solver_impl.h"
#include "icp_device.h"
#include "icp_kernels.h"

namespace icpgpu {
namespace {

constexpr int GK = kGicpK;  // 20

__device__ __forceinline__ bool finite3g(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

__device__ __forceinline__ void cell_of_g(const GridDesc& g, float x, float y, float z, int& cx, int& cy, int& cz) {
  const float fx = floorf((x - g.ox) * g.inv_h), fy = floorf((y - g.oy) * g.inv_h), fz = floorf((z - g.oz) * g.inv_h);
  const float lim = 1048576.0f;
  cx = (int)fminf(fmaxf(fx, -lim), lim);
  cy = (int)fminf(fmaxf(fy, -lim), lim);
  cz = (int)fminf(fmaxf(fz, -lim), lim);
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {  // lane: wave-uniform
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Left singular vectors of a symmetric 3x3 (row-major), columns by decreasing singular value: Eigen::JacobiSVD<Matrix3d>(A,
// ComputeFullU).matrixU() as the CPU checker restates it for computeCovariances (Eigen 3.3's two-sided Jacobi iteration on the
// scaled matrix, pairs (1,0) (2,0) (2,1), a pair rotated while an off-diagonal exceeds 2 eps x the largest diagonal met so far,
// real_2x2_jacobi_svd + makeJacobi, diagonal signs into U, descending selection sort) -- OPERATION FOR OPERATION: on a patch
// with two (nearly) equal singular values the vectors are whatever the method's rounding makes them, and a regularised
// covariance that differs in its 10th digit is a different optimisation problem for the chaotic BFGS that follows.
// Until round 3 both sides ran a one-sided iteration whose stopping rule was relative to each column pair: 0.6 % of a scan's
// patches (near-planar: the third column is noise) hovered at it for all 60 sweeps and every wave waited for such a lane
// (136 us per 22k-point cloud); Eigen's rule is relative to the LARGEST singular value and ends after 2-4 sweeps.
constexpr int kSvdMaxSweeps = 64;  // (Eigen has no cap; this one only guards non-finite input, the checker has the same)
__device__ __forceinline__ void svd_rot_rows(double* W, int p, int q, double c, double s) {
  if (c == 1.0 && s == 0.0) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double xi = W[3 * p + i], yi = W[3 * q + i];
    W[3 * p + i] = c * xi + s * yi;
    W[3 * q + i] = -s * xi + c * yi;
  }
}
__device__ __forceinline__ void svd_rot_cols(double* W, int p, int q, double c, double s) {
  if (c == 1.0 && s == 0.0) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double xi = W[3 * i + p], yi = W[3 * i + q];
    W[3 * i + p] = c * xi - s * yi;
    W[3 * i + q] = s * xi + c * yi;
  }
}
__device__ void svd3_left_vectors(const double A[9], double U[9]) {
  const double kMin = 2.2250738585072014e-308, kEps = 2.220446049250313e-16;  // DBL_MIN, DBL_EPSILON
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    if (fabs(A[i]) > scale) scale = fabs(A[i]);
  if (scale == 0.0) scale = 1.0;
  double W[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    W[i] = A[i] / scale;
    U[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  const double precision = 2.0 * kEps;
  double max_diag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
  for (int sweep = 0; sweep < kSvdMaxSweeps; ++sweep) {
    bool finished = true;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 0 ? 1 : 2, q = pq == 2 ? 1 : 0;  // (1,0) (2,0) (2,1)
      const double threshold = fmax(kMin, precision * max_diag);
      if (!(fabs(W[3 * p + q]) > threshold || fabs(W[3 * q + p]) > threshold)) continue;
      finished = false;
      double m00 = W[3 * p + p], m01 = W[3 * p + q], m10 = W[3 * q + p], m11 = W[3 * q + q];
      double c1, s1;
      const double t = m00 + m11, d = m10 - m01;
      if (fabs(d) < kMin) {
        s1 = 0.0;
        c1 = 1.0;
      } else {
        const double u = t / d, tmp = sqrt(1.0 + u * u);
        s1 = 1.0 / tmp;
        c1 = u / tmp;
      }
      if (!(c1 == 1.0 && s1 == 0.0)) {
        const double a0 = m00, a1 = m01, b0 = m10, b1 = m11;
        m00 = c1 * a0 + s1 * b0;
        m01 = c1 * a1 + s1 * b1;
        m10 = -s1 * a0 + c1 * b0;
        m11 = -s1 * a1 + c1 * b1;
      }
      double cr, sr;
      const double deno = 2.0 * fabs(m01);
      if (deno < kMin) {
        cr = 1.0;
        sr = 0.0;
      } else {
        const double tau = (m00 - m11) / deno, w = sqrt(tau * tau + 1.0);
        const double tt = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
        const double sign_t = tt > 0.0 ? 1.0 : -1.0, n = 1.0 / sqrt(tt * tt + 1.0);
        sr = -sign_t * (m01 / fabs(m01)) * fabs(tt) * n;
        cr = n;
      }
      const double cl = c1 * cr - s1 * (-sr), sl = c1 * (-sr) + s1 * cr;
      svd_rot_rows(W, p, q, cl, sl);
      svd_rot_cols(U, p, q, cl, -sl);
      svd_rot_cols(W, p, q, cr, sr);
      max_diag = fmax(max_diag, fmax(fabs(W[3 * p + p]), fabs(W[3 * q + q])));
    }
    if (finished) break;
  }
  double sv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a = fabs(W[4 * i]);
    sv[i] = a * scale;
    if (a != 0.0) {
      const double f = W[4 * i] / a;
#pragma unroll
      for (int r = 0; r < 3; ++r) U[3 * r + i] *= f;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int pos = i;
#pragma unroll
    for (int k = i + 1; k < 3; ++k)
      if (sv[k] > sv[pos]) pos = k;
    if (sv[pos] == 0.0) break;
    if (pos != i) {
      const double ts = sv[i];
      sv[i] = sv[pos];
      sv[pos] = ts;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double tu = U[3 * r + i];
        U[3 * r + i] = U[3 * r + pos];
        U[3 * r + pos] = tu;
      }
    }
  }
}

// ---- computeCovariances -----------------------------------------------------------------------------------------
// (list, optional: the points gicp_cov_select_kernel left over -- list[0 .. *list_n); without it every point of the cloud)
__global__ __launch_bounds__(256) void gicp_cov_kernel(const float4* __restrict__ cloud, int n,
                                                       const float4* __restrict__ sorted,
                                                       const int* __restrict__ cell_start, GridDesc g,
                                                       double* __restrict__ cov6, const int* __restrict__ list,
                                                       const int* __restrict__ list_n) {
  __shared__ unsigned long long s_key[4][GK];  // per-wave staging of the top-20 while it is re-ranked
  __shared__ int s_pos[4][GK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // one wave per point; with a list: a few hundred waves walk it (it is short or empty: an empty launch sized for the cloud cost 4.6 us)
  for (int slot = blockIdx.x * 4 + wv;; slot += gridDim.x * 4) {
  int i = slot;
  if (list) {
    if (slot >= *list_n) return;
    i = list[slot];
  }
  if (i >= n) return;
  const float4 s = cloud[i];
  double C[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 1.0};
  bool have_patch = false;
  if (finite3g(s.x, s.y, s.z)) {
    int cx, cy, cz;
    cell_of_g(g, s.x, s.y, s.z, cx, cy, cz);
    const int span = max(g.nx, max(g.ny, g.nz));
    unsigned long long mykey = kEmptyKey;  // lanes 0..19: the running top-20 (unordered)
    int mypos = -1;
    for (int rho = 1;; rho *= 2) {
      mykey = kEmptyKey;
      mypos = -1;
      unsigned long long kth = kEmptyKey;  // largest key among the 20 slots
      const int side = 2 * rho + 1, nrows = side * side;
      const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.nx - 1);
      const float inv_side = 1.0f / (float)side;
      for (int rb = 0; rb < nrows; rb += 64) {
        const int r = rb + lane;
        const int zr = (int)(((float)r + 0.5f) * inv_side), yr = r - zr * side;
        const int yy = cy + yr - rho, zz = cz + zr - rho;
        int lo = 0, len = 0;
        if (r < nrows && x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
          const int row = zz * g.sz + yy * g.sy;
          lo = cell_start[row + x0];
          len = cell_start[row + x1 + 1] - lo;
        }
        // The batch's rows as ONE list of candidates, 64 per chunk (round 4; until then one row -- ~24 candidates at cells sized for
        // ~8 points then -- per chunk, later two): an exclusive scan of the row lengths over the lanes, and every lane of a chunk finds
        // the row of its entry by a binary search over those offsets (six lane reads).  A merge round's ~100 fixed instructions
        // are then spent on 64 candidates, not on 24.
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d, 64);
          if ((int)lane >= d) incl += t;
        }
        const int start = incl - len, total = __shfl(incl, 63, 64);
        for (int c0 = 0; c0 < total; c0 += 64) {  // wave-uniform
          {
            const int e = c0 + lane;
            int rr = 0;  // the LAST lane whose offset is <= e (empty rows share their successor's offset: the successor wins)
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
              const int probe = rr + step;
              const int sp = __shfl(start, probe & 63, 64);
              if (probe < 64 && sp <= e) rr = probe;
            }
            const int rlo = __shfl(lo, rr, 64), rst = __shfl(start, rr, 64);
            unsigned long long ckey = kEmptyKey;
            const int cpos = rlo + (e - rst);
            if (e < total) {
              const float4 q = sorted[cpos];
              const float d = dist2(q.x, q.y, q.z, s.x, s.y, s.z);
              ckey = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q.w);
            }
            const unsigned long long better = __ballot(ckey < kth);
            if (better) {  // wave-uniform
              // Merge the chunk's improving candidates into the sorted top-20 (lanes 0..19) by RANK: keys are distinct
              // (the index is part of the key), so "how many keys of the union are smaller than mine" is my slot.
              // ~80 + 6 * popcount(better) instructions per chunk, against ~40 per candidate for one-at-a-time insertion
              // (measured: covariances of a 200k-point cloud 4.75 -> 4.25 ms; most of the kernel's time is the row walk).
              int rank_top = (int)lane;  // slots below me among the current top keys (they are sorted); lanes < GK only
              int rank_cand = 0;
              for (unsigned long long m = better; m; m &= m - 1) {
                const int j = __ffsll((long long)m) - 1;
                const unsigned long long bk = readlane_u64(ckey, j);
                rank_top += bk < mykey ? 1 : 0;
                rank_cand += bk < ckey ? 1 : 0;
              }
#pragma unroll
              for (int j = 0; j < GK; ++j) rank_cand += readlane_u64(mykey, j) < ckey ? 1 : 0;  // empty slots hold the maximum key
              if (lane < GK) {
                s_key[wv][lane] = kEmptyKey;
                s_pos[wv][lane] = -1;
              }
              if (lane < GK && mykey != kEmptyKey && rank_top < GK) {
                s_key[wv][rank_top] = mykey;
                s_pos[wv][rank_top] = mypos;
              }
              if (((better >> lane) & 1ull) && rank_cand < GK) {
                s_key[wv][rank_cand] = ckey;
                s_pos[wv][rank_cand] = cpos;
              }
              if (lane < GK) {
                mykey = s_key[wv][lane];
                mypos = s_pos[wv][lane];
              }
              kth = readlane_u64(mykey, GK - 1);
            }
          }
        }
      }
      const float safe = (float)rho * g.h * kGridSafety;
      const bool full = kth != kEmptyKey;  // all 20 slots filled
      if ((full && __uint_as_float((unsigned int)(kth >> 32)) <= safe * safe) || rho >= span) break;
    }
    // lanes 0..19 hold the neighbours: mean and second moments (float products, double sums), wave-reduced
    double sx = 0, sy = 0, sz = 0, xx = 0, yx = 0, yy2 = 0, zx = 0, zy = 0, zz2 = 0;
    int have = 0;
    if (lane < GK && mypos >= 0) {
      const float4 q = sorted[mypos];
      sx = q.x; sy = q.y; sz = q.z;
      xx = (double)(q.x * q.x);
      yx = (double)(q.y * q.x);
      yy2 = (double)(q.y * q.y);
      zx = (double)(q.z * q.x);
      zy = (double)(q.z * q.y);
      zz2 = (double)(q.z * q.z);
      have = 1;
    }
    const int cnt = __popcll(__ballot(have));
    sx = wave_sum_d(sx); sy = wave_sum_d(sy); sz = wave_sum_d(sz);
    xx = wave_sum_d(xx); yx = wave_sum_d(yx); yy2 = wave_sum_d(yy2);
    zx = wave_sum_d(zx); zy = wave_sum_d(zy); zz2 = wave_sum_d(zz2);
    if (cnt == GK) {  // hand the sample covariance (6 distinct entries) to gicp_cov_finish_kernel
      const double k = (double)GK;
      const double mx = sx / k, my = sy / k, mz = sz / k;
      C[0] = xx / k - mx * mx;
      C[1] = yx / k - my * mx;
      C[2] = zx / k - mz * mx;
      C[3] = yy2 / k - my * my;
      C[4] = zy / k - mz * my;
      C[5] = zz2 / k - mz * mz;
      have_patch = true;
    }
  }
  if (!have_patch) C[0] = __longlong_as_double(0x7FF8000000000000ll);  // marker: identity covariance
  if (lane < 6) cov6[(size_t)i * 6 + lane] = C[lane];
  if (!list) return;
  }
}

// ---- shared by the selecting kernels: from a list of >= 20 keys to the covariance ---------------------------------------------
constexpr int CS = 4;            // 256 keys per level in registers (a scan filtered at 0.2 m has ~60 inside a level's radius)
constexpr int CS_LIST = 64 * CS;

// Lanes of ONE wave hand data to each other through LDS below (a list appended to by some lanes and read by others).  The LDS
// operations of a wave complete in order, so no instruction is needed -- but the memory model orders one lane's stores before
// another lane's loads only across a synchronisation: a wavefront-scope release / acquire fence pair around a wave barrier says so
// (no instruction beyond, at most, a wait for the LDS counter).
__device__ __forceinline__ void wave_lds_handover() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// buf[0 .. fill): distinct keys (d2 bits << 32 | index), all <= cap_key, 20 <= fill <= CS_LIST; one wave.  On return lanes 0..19
// of `top` (LDS, 20 words) hold the 20 smallest in ascending order.  The threshold is lowered on the float distance while that
// separates, then by bisection on the 64-bit keys themselves (dozens of equal distances: the index part decides) -- keys are
// distinct, so a threshold with 20..64 keys below it always exists.
__device__ __forceinline__ void cov_pick20(unsigned long long* buf, unsigned long long* top, int fill, unsigned long long cap_key, int lane,
                                           unsigned long long lane_lt, int& probes, int& ranked) {
  unsigned long long K[CS];
  const int nreg = (fill + 63) >> 6;  // wave-uniform: registers in use
  wave_lds_handover();  // (the list was appended to by whichever lanes found the keys)
#pragma unroll
  for (int m = 0; m < CS; ++m) K[m] = (m < nreg && m * 64 + lane < fill) ? buf[m * 64 + lane] : kEmptyKey;
  auto count_le = [&](unsigned long long T) {
    int c = 0;
#pragma unroll
    for (int m = 0; m < CS; ++m)
      if (m < nreg) c += __popcll(__ballot(K[m] <= T));
    return c;
  };
  // lower the threshold until 20 .. 64 keys pass: [T_lo: fewer than 20 (at most the key 0 itself), T: at least 20]
  unsigned long long T_lo = 0ull, T = cap_key;
  int c_lo = 0, c_hi = fill;
  bool by_key = false;
  for (int probe = 0; c_hi > 64; ++probe) {
    unsigned long long Tm = 0ull;
    if (!by_key) {
      // interpolate for ~40 on d2 (the first probes), bisect afterwards
      const float lo_f = __uint_as_float((unsigned int)(T_lo >> 32)), hi_f = __uint_as_float((unsigned int)(T >> 32));
      float mid = probe < 6 ? lo_f + (hi_f - lo_f) * ((40.0f - (float)c_lo) / (float)(c_hi - c_lo)) : 0.5f * (lo_f + hi_f);
      if (!(mid > lo_f && mid < hi_f)) mid = 0.5f * (lo_f + hi_f);
      if (probe >= 12 || !(mid > lo_f && mid < hi_f)) by_key = true;  // (adjacent floats: more than 44 keys at one distance)
      else Tm = ((unsigned long long)__float_as_uint(mid) << 32) | 0xFFFFFFFFull;
    }
    if (by_key) Tm = T_lo + ((T - T_lo) >> 1);  // strictly inside: T - T_lo >= 2 while more than one key lies between them
    probes += 1;
    const int cm = count_le(Tm);
    if (cm < GK) {
      T_lo = Tm;
      c_lo = cm;
    } else {
      T = Tm;
      c_hi = cm;
    }
  }
  // compact the passing keys, one per lane (the list is in registers: its first 64 entries are free again)
  unsigned long long k = K[0];  // (a list of one register's length that passes whole is in place already)
  if (!(nreg == 1 && c_hi == fill)) {
    wave_lds_handover();  // (every lane has its part of the list in registers before the buffer is written again)
    int base = 0;
#pragma unroll
    for (int m = 0; m < CS; ++m)
      if (m < nreg) {
        const unsigned long long b = __ballot(K[m] <= T);
        if (K[m] <= T) buf[base + __popcll(b & lane_lt)] = K[m];
        base += __popcll(b);
      }
    wave_lds_handover();
    k = lane < c_hi ? buf[lane] : kEmptyKey;
  }
  int rank = 0;
  const int c8 = c_hi & ~7;
  for (int j = 0; j < c8; j += 8) {  // wave-uniform addresses: broadcast reads, eight in flight
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = buf[j + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) rank += v[u] < k ? 1 : 0;
  }
  for (int j = c8; j < c_hi; ++j) rank += buf[j] < k ? 1 : 0;
  if (lane < c_hi && rank < GK) top[rank] = k;
  wave_lds_handover();  // (top[] is read by lanes 0..19 next: cov_emit)
  ranked = c_hi;
}

// Lanes 0..19 of `top` hold the neighbours in key order: mean and second moments exactly as gicp_cov_kernel forms them -- there nine
// butterfly sums over the wave (xor 32, 16, 8, 4, 2, 1 with zeros in the lanes above 19) and nine divisions in every lane; here
// the SAME tree of additions (fp addition commutes, so the butterfly's value in lane 0 is a fixed tree over the 20 terms: the xor-32
// step adds a zero, the xor-16 step pairs terms l and l + 16 for l < 4 and adds zeros elsewhere, then 8 + 4 + 2 + 1 pairings) is
// walked by ONE lane per sum over the terms in LDS, and one lane per covariance entry does its three divisions.  One wave;
// `scratch`: 189 doubles of LDS (the key list's buffer: the list is dead).
__device__ __forceinline__ void cov_emit(const unsigned long long* top, double* scratch, const float4* __restrict__ cloud, int i,
                                         double* __restrict__ cov6, int lane) {
  double* term = scratch;  // [9][20]
  if (lane < GK) {
    const float4 q = cloud[(unsigned int)top[lane]];  // (the grid's sorted copy holds the same coordinates under the same index)
    term[0 * GK + lane] = (double)q.x;
    term[1 * GK + lane] = (double)q.y;
    term[2 * GK + lane] = (double)q.z;
    term[3 * GK + lane] = (double)(q.x * q.x);  // (the second moments in the order of the triangle's entries)
    term[4 * GK + lane] = (double)(q.y * q.x);
    term[5 * GK + lane] = (double)(q.z * q.x);
    term[6 * GK + lane] = (double)(q.y * q.y);
    term[7 * GK + lane] = (double)(q.z * q.y);
    term[8 * GK + lane] = (double)(q.z * q.z);
  }
  double* sums = term + 9 * GK;  // [9]
  wave_lds_handover();
  if (lane < 9) {
    const volatile double* v = term + lane * GK;  // (volatile: read in the order of use -- all twenty at once cost 40 registers)
    // a(l) = lane l's value after the xor-32 and xor-16 steps; then c(l) = a(l) + a(l + 8), d(l) = c(l) + c(l + 4),
    // e(0) = d(0) + d(2), e(1) = d(1) + d(3), sum = e(0) + e(1)
    auto a = [&](int l) -> double { return (v[l] + 0.0) + (l < 4 ? v[l + 16] + 0.0 : 0.0); };
    auto d = [&](int l) -> double { return (a(l) + a(l + 8)) + (a(l + 4) + a(l + 12)); };
    const double e0 = d(0) + d(2);
    const double e1 = d(1) + d(3);
    sums[lane] = e0 + e1;
  }
  wave_lds_handover();
  if (lane < 6) {
    // entry e = (row r, column c) of the upper triangle: (0,0) (1,0) (2,0) (1,1) (2,1) (2,2), second moments at sums[3 + e]
    const int r = lane == 0 ? 0 : (lane == 1 || lane == 3 ? 1 : 2), cidx = lane < 3 ? 0 : (lane < 5 ? 1 : 2);
    const double kk = (double)GK;
    const double mr = sums[r] / kk, mc = sums[cidx] / kk;
    cov6[(size_t)i * 6 + lane] = sums[3 + lane] / kk - mr * mc;
  }
}

// The points the selecting kernel's cubes do not reach cheaply -- isolated far-field points whose 20th neighbour is metres away: cube
// radii 8, 16, ... cells are hundreds to thousands of rows walked in one dependent chain (27 row batches, ~65 us, for a point
// certified at radius 16; a handful of them per scan set the kernel's duration) -- and the rare ones its caps cannot settle
// (hundreds of points at one distance).  Here a WORKGROUP takes such a point and looks at the whole cloud (a filtered scan: ~23k
// points, 23 per thread): a pass collects the keys under a cap into the list; the first cap extrapolates the count the last cube
// found inside its radius (neighbours on a surface grow with d2) to ~60, later ones interpolate between the caps tried -- usually
// one pass -- and when float distances no longer separate, the caps become 64-bit keys and are bisected as such (keys are distinct:
// that always ends).  Then the same selection and sums as above.  Exact whatever the caps were: every point of the cloud is looked
// at; a cloud with fewer than 20 finite points gets the identity marker, as in gicp_cov_kernel.  It is also the whole covariance
// pass of a cloud the k-NN grid refuses (thousands of points in one cell of the largest table: gicp_cov_all_far_kernel queues every
// point).  Clouds above kGicpCovFarMost points keep the streaming kernel for these points.
// far_list entries: point index | code << 20; code 0..19 = the keys the last cube found inside r_tried, 31 = no such knowledge.
constexpr int kCovFarLevel = 4;  // the last cube radius (cells) the selecting kernel tries before it hands a point over
constexpr int kCovFarNoHint = 31;
__global__ __launch_bounds__(1024) void gicp_cov_far_kernel(const float4* __restrict__ cloud, int n, double* __restrict__ cov6, float r_tried,
                                                            const int* __restrict__ far_list, const int* __restrict__ far_n) {
  __shared__ unsigned long long s_buf[CS_LIST];
  __shared__ unsigned long long s_top[GK];
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned long long lane_lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  auto key_of = [](float d2, unsigned int idx) -> unsigned long long { return ((unsigned long long)__float_as_uint(d2) << 32) | idx; };
  for (int slot = blockIdx.x; slot < *far_n; slot += gridDim.x) {
    const int packed = far_list[slot];
    const int i = packed & 0xFFFFF, code = (packed >> 20) & 31;
    const float4 s = cloud[i];
    // the keys <= cap: their number, the first CS_LIST of them in s_buf (any order: they are ranked later)
    auto pass = [&](unsigned long long cap) -> int {
      __syncthreads();
      if (tid == 0) s_count = 0;
      __syncthreads();
      for (int j0 = 0; j0 < n; j0 += 8192) {  // block-uniform; eight loads in flight per thread
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * 1024 + tid;
          q[u] = j < n ? cloud[j] : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * 1024 + tid;
          const unsigned long long key = key_of(dist2(q[u].x, q[u].y, q[u].z, s.x, s.y, s.z), (unsigned int)j);
          const bool in = finite3g(q[u].x, q[u].y, q[u].z) && key <= cap;  // (the grid's sorted copy leaves non-finite points out: so do we)
          const unsigned long long b = __ballot(in);
          if (b) {  // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_count, __popcll(b));
            base = __shfl(base, 0, 64);
            const int at = base + __popcll(b & lane_lt);
            if (in && at < CS_LIST) s_buf[at] = key;
          }
        }
      }
      __syncthreads();
      return s_count;
    };
    // [lo: fewer than 20 keys (the key 0 alone at most), hi: more than the list holds]
    const bool hint = code != kCovFarNoHint;
    unsigned long long lo = hint ? key_of(r_tried * r_tried, 0xFFFFFFFFu) : 0ull, hi = 0ull;
    int n_lo = hint ? code : 0, n_hi = 0, fill = 0;
    unsigned long long cap = key_of(r_tried * r_tried * (hint ? 60.0f / (float)max(code, 2) : 1.0f), 0xFFFFFFFFu);
    bool settled = false, by_key = false;
    for (int pass_no = 0; pass_no < 200; ++pass_no) {
      fill = pass(cap);
      if (fill >= GK && fill <= CS_LIST) {
        settled = true;
        break;
      }
      if (fill < GK) {
        lo = cap;
        n_lo = fill;
      } else {
        hi = cap;
        n_hi = fill;
      }
      if (n_hi == 0) {  // nothing above yet: grow (by the same rule)
        const float c = __uint_as_float((unsigned int)(cap >> 32));
        float next = c * (fill < 5 ? 8.0f : 60.0f / (float)fill);
        if (!(next > c)) next = c > 0.0f ? c * 2.0f : 1.0e-12f;
        if (!(next < 3.0e38f)) break;  // fewer than 20 finite points in the cloud
        cap = key_of(next, 0xFFFFFFFFu);
        continue;
      }
      if (!by_key) {
        const float lo_f = __uint_as_float((unsigned int)(lo >> 32)), hi_f = __uint_as_float((unsigned int)(hi >> 32));
        float mid = pass_no < 8 ? lo_f + (hi_f - lo_f) * ((60.0f - (float)n_lo) / (float)(n_hi - n_lo)) : 0.5f * (lo_f + hi_f);
        if (!(mid > lo_f && mid < hi_f)) mid = 0.5f * (lo_f + hi_f);
        if (pass_no >= 40 || !(mid > lo_f && mid < hi_f)) by_key = true;  // hundreds of keys at one distance: the index part decides
        else cap = key_of(mid, 0xFFFFFFFFu);
      }
      if (by_key) cap = lo + ((hi - lo) >> 1);
    }
    if (tid < 64) {  // one wave finishes
      if (settled) {
        int probes = 0, ranked = 0;
        cov_pick20(s_buf, s_top, fill, cap, lane, lane_lt, probes, ranked);
        cov_emit(s_top, reinterpret_cast<double*>(s_buf), cloud, i, cov6, lane);
      } else if (lane < 6) {  // (the marker of an identity covariance, as gicp_cov_kernel writes it for a cloud of fewer than 20 points)
        cov6[(size_t)i * 6 + lane] = lane == 0 ? __longlong_as_double(0x7FF8000000000000ll) : (lane == 3 || lane == 5 ? 1.0 : 0.0);
      }
    }
  }
}

// every finite point of the cloud onto the far-field kernel's list (no hint), the marker for the others: the covariance pass of a
// cloud that has no grid
__global__ __launch_bounds__(256) void gicp_cov_all_far_kernel(const float4* __restrict__ cloud, int n, double* __restrict__ cov6,
                                                               int* __restrict__ far_list, int* __restrict__ far_n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 s = cloud[i];
  if (finite3g(s.x, s.y, s.z)) {
    far_list[atomicAdd(far_n, 1)] = i | (kCovFarNoHint << 20);
  } else {
    double* c = cov6 + (size_t)i * 6;
    c[0] = __longlong_as_double(0x7FF8000000000000ll);
    c[1] = c[2] = c[4] = 0.0;
    c[3] = c[5] = 1.0;
  }
}

// computeCovariances, the SELECTING form (round 6).  gicp_cov_kernel above streams candidates 64 at a time into a running top-20 and
// pays ~100 + 6 x (improving candidates) instructions per chunk -- 460 for the first chunk of every level, when all 64 improve --
// in ONE dependent chain per point; a dense corner of a scan (a thousand candidates in the first cube) kept a wave for 30 us.
// Here a level's candidates (the cube of radius rho cells around the point) are only EVALUATED, and the keys (d2 bits << 32 |
// index) that lie within a distance cap -- the level's certified radius to begin with -- are appended to a per-wave LDS list:
//   * a level succeeds iff >= 20 keys lie within its certified radius (the 20th neighbour is then known to be the true one); a
//     level that fails has cost nothing but the evaluations;
//   * a list that overflows (> 256 keys inside the radius: a dense corner) is collected again under a smaller cap, aimed at ~128
//     keys by interpolating on d2 (neighbours on a surface grow linearly with it);
//   * from the list, held in registers, a threshold is lowered the same way until 20..64 keys pass; those are compacted into one
//     lane each and ranked by counting (keys are distinct): ranks 0..19 are the neighbours, in key order.
// The next chunk's load is in flight while a chunk is processed, and rows are found by counting over at most 25 lane reads instead
// of a six-step search through the LDS crossbar: the kernel is a latency chain per point, not an issue problem (23k waves for 8k
// slots).  Same neighbours, same order, same sums as gicp_cov_kernel (tests/test_gpu_gicp.py compares the covariances bit for bit
// with it and with the oracle).  A point whose caps do not settle (dozens of equal distances) or whose search reaches the whole
// grid goes on a list, and gicp_cov_kernel finishes the list.
template <bool STATS>
#if defined(ICPGPU_COV_WAVES8)  // A/B build: 64 registers (a few spilled) for 8 waves per SIMD instead of 72 for 7 (no difference measured)
__attribute__((amdgpu_waves_per_eu(8, 8)))
#else
__attribute__((amdgpu_waves_per_eu(7, 8)))  // (left alone the compiler spreads to 108 registers: 4 waves per SIMD for a latency-bound kernel)
#endif
__global__ __launch_bounds__(256) void gicp_cov_select_kernel(const float4* __restrict__ cloud, int n,
                                                              const float4* __restrict__ sorted,
                                                              const int* __restrict__ cell_start, GridDesc g,
                                                              double* __restrict__ cov6, int* __restrict__ list,
                                                              int* __restrict__ list_n, int far_ok, int* __restrict__ far_list,
                                                              int* __restrict__ far_n, unsigned long long* __restrict__ stats) {
#define CS_STAT(k, v) do { if (STATS && lane == 0) atomicAdd(stats + (k), (unsigned long long)(v)); } while (0)
  __shared__ unsigned long long s_buf[4][CS_LIST];
  __shared__ unsigned long long s_top[4][GK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv;  // one wave per point
  if (i >= n) return;
  const float4 s = cloud[i];
  if (!finite3g(s.x, s.y, s.z)) {  // (as gicp_cov_kernel: the marker of an identity covariance)
    if (lane < 6) cov6[(size_t)i * 6 + lane] = lane == 0 ? __longlong_as_double(0x7FF8000000000000ll) : (lane == 3 || lane == 5 ? 1.0 : 0.0);
    return;
  }
  int cx, cy, cz;
  cell_of_g(g, s.x, s.y, s.z, cx, cy, cz);
  const int span = max(g.nx, max(g.ny, g.nz));
  const unsigned long long lane_lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  unsigned long long* buf = s_buf[wv];
  bool done = false, give_up = false, go_far = false;
  int last_seen = 0;  // keys inside the radius of the last level that failed
  for (int rho = 1; !done && !give_up; rho *= 2) {
    if (far_ok && rho > kCovFarLevel) {  // wave-uniform
      go_far = true;
      break;
    }
    const int side = 2 * rho + 1, nrows = side * side;
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.nx - 1);
    const float inv_side = 1.0f / (float)side;
    const float safe = (float)rho * g.h * kGridSafety;
    const float safe_sq = safe * safe;
    // One pass over the level's candidates: the keys with d2 <= cap into buf (the first CS_LIST of them), their number returned.
    // Under a cap below the level's radius only the cells the cap's ball reaches are read (box pruning on the SAME float cell
    // coordinates that binned the points, which are monotone in x: a point within sqrt(cap) of s cannot lie in a skipped cell).
    // A list that overflows ends the pass at once (levels of one row batch): `est` then extrapolates the count from the share of
    // the candidates seen so far -- a guide for the next cap, not a result.
    auto collect = [&](float cap, int& est) -> int {
      int fill = 0;
      const float r_cap = sqrtf(cap) * 1.0001f + g.h * 0.03125f;
      const int xa = (int)fmaxf(floorf((s.x - r_cap - g.ox) * g.inv_h), -4.0f), xb = (int)fminf(floorf((s.x + r_cap - g.ox) * g.inv_h), 1048576.0f);
      const int ya = (int)fmaxf(floorf((s.y - r_cap - g.oy) * g.inv_h), -4.0f), yb = (int)fminf(floorf((s.y + r_cap - g.oy) * g.inv_h), 1048576.0f);
      const int za = (int)fmaxf(floorf((s.z - r_cap - g.oz) * g.inv_h), -4.0f), zb = (int)fminf(floorf((s.z + r_cap - g.oz) * g.inv_h), 1048576.0f);
      const int xx0 = max(x0, xa), xx1 = min(x1, xb);
      for (int rb = 0; rb < nrows; rb += 64) {
        const int r = rb + lane;
        const int zr = (int)(((float)r + 0.5f) * inv_side), yr = r - zr * side;
        const int yy = cy + yr - rho, zz = cz + zr - rho;
        int lo = 0, len = 0;
        if (r < nrows && xx0 <= xx1 && yy >= max(0, ya) && yy <= min(g.ny - 1, yb) && zz >= max(0, za) && zz <= min(g.nz - 1, zb)) {
          const int row = zz * g.sz + yy * g.sy;
          lo = cell_start[row + xx0];
          len = cell_start[row + xx1 + 1] - lo;
        }
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d, 64);
          if ((int)lane >= d) incl += t;
        }
        const int start = incl - len, total = __shfl(incl, 63, 64);
        if (nrows <= 64 && total < GK) {  // fewer than 20 points in reach: nothing to evaluate
          est = total;
          return total;
        }
        const int rows_here = min(nrows - rb, 64);
        // entry e of the batch's candidate list lives in the LAST row whose offset is <= e (empty rows share their successor's
        // offset: the successor wins) -- (rows with offset <= e) - 1, counted over lane reads when the rows are few
        auto locate = [&](int e) -> int {
          int rr = 0;
          if (rows_here <= 25) {
            for (int j = 1; j < rows_here; ++j) rr += __builtin_amdgcn_readlane(start, j) <= e ? 1 : 0;
          } else {
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
              const int probe = rr + step;
              const int sp = __shfl(start, probe & 63, 64);
              if (probe < 64 && sp <= e) rr = probe;
            }
          }
          return __shfl(lo, rr, 64) + (e - __shfl(start, rr, 64));
        };
        // (locate() moves data between lanes: every lane of the wave takes part in it, only the load is conditional)
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        int at0 = locate(lane);
        if (lane < total) q = sorted[at0];
        for (int c0 = 0; c0 < total; c0 += 64) {  // wave-uniform
          const float4 cur = q;
          const bool live = c0 + lane < total;
          if (c0 + 64 < total) {  // wave-uniform: the next chunk's load, in flight during this one
            const int at1 = locate(c0 + 64 + lane);
            if (c0 + 64 + lane < total) q = sorted[at1];
          }
          const float d = dist2(cur.x, cur.y, cur.z, s.x, s.y, s.z);
          const bool pass = live && d <= cap;
          const unsigned long long b = __ballot(pass);
          const int at = fill + __popcll(b & lane_lt);
          if (pass && at < CS_LIST) buf[at] = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(cur.w);
          fill += __popcll(b);
          if (fill > CS_LIST && nrows <= 64) {  // wave-uniform: overflow
            const int seen = min(c0 + 64, total);
            est = (int)fminf((float)fill * ((float)total / (float)seen), 1.0e9f);
            return fill;
          }
        }
      }
      est = fill;
      return fill;
    };
    CS_STAT(2, 1);
    // the level's radius first; a list that overflows (a dense corner) again under smaller caps, [lo_c: fewer than 20 keys, hi_c: more
    // than the list holds], aimed at ~100 keys by interpolating on d2
    float cap = safe_sq, lo_c = 0.0f, hi_c = safe_sq;
    int fill = 0, est = 0, n_lo = 0, n_hi = 0;
    bool level_fails = false;
    for (int pass_no = 0;; ++pass_no) {
      fill = collect(cap, est);
      if (pass_no == 0 && fill < GK) {  // the level cannot certify 20 neighbours
        level_fails = true;
        break;
      }
      if (fill >= GK && fill <= CS_LIST) break;
      if (pass_no == 0) CS_STAT(3, 1);
      if (fill < GK) {
        lo_c = cap;
        n_lo = fill;
      } else {
        hi_c = cap;
        n_hi = est;
      }
      float mid = pass_no < 4 ? lo_c + (hi_c - lo_c) * ((100.0f - (float)n_lo) / (float)(n_hi - n_lo)) : 0.5f * (lo_c + hi_c);
      if (!(mid > lo_c && mid < hi_c)) mid = 0.5f * (lo_c + hi_c);
      if (pass_no >= 12 || !(mid > lo_c && mid < hi_c)) {
        give_up = true;
        CS_STAT(4, 1);
        break;
      }
      cap = mid;
    }
    if (level_fails) {
      last_seen = fill;
      if (rho >= span) { give_up = true; CS_STAT(5, 1); }  // (the whole grid has been looked at: gicp_cov_kernel's closing rule applies)
      continue;
    }
    if (give_up) break;
    CS_STAT(1, fill);
    int probes = 0, ranked = 0;
    cov_pick20(buf, s_top[wv], fill, ((unsigned long long)__float_as_uint(cap) << 32) | 0xFFFFFFFFull, lane, lane_lt, probes, ranked);
    done = true;
    CS_STAT(6, probes);
    CS_STAT(7, ranked);
    CS_STAT(0, 1);
    CS_STAT(8 + min(31 - __clz(rho), 7), 1);  // histogram of the level that succeeded: rho = 1, 2, 4, ... 128+
  }
#undef CS_STAT
  if (!done) {
    // far_ok (clouds up to kGicpCovFarMost points): whatever is not done -- no cube of up to kCovFarLevel cells certified the point,
    // the caps did not settle -- goes to gicp_cov_far_kernel; in larger clouds to gicp_cov_kernel
    if (lane == 0) {
      if (go_far) far_list[atomicAdd(far_n, 1)] = i | (last_seen << 20);  // (i < kGicpCovFarMost, last_seen < 20)
      else if (far_ok) far_list[atomicAdd(far_n, 1)] = i | (kCovFarNoHint << 20);  // caps that did not settle, the whole grid searched
      else list[atomicAdd(list_n, 1)] = i;
    }
    return;
  }
  cov_emit(s_top[wv], reinterpret_cast<double*>(buf), cloud, i, cov6, lane);
}

// The 3x3 decomposition, ONE LANE PER POINT: inside the search kernel every lane of the wave repeated it (several thousand
// double-precision instructions per point, the largest part of that kernel's time); here 64 points share a wave.
// (zero2, optional: the two list counters of launch_gicp_covariances, left zero for the next cloud -- every kernel that reads them has run)
__global__ __launch_bounds__(256) void gicp_cov_finish_kernel(int n, double* __restrict__ cov6, int* __restrict__ zero2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (zero2 && i == 0) zero2[0] = zero2[1] = 0;
  if (i >= n) return;
  double* c = cov6 + (size_t)i * 6;
  const double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3], a4 = c[4], a5 = c[5];
  double C[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 1.0};
  if (a0 == a0) {  // not the marker
    const double A[9] = {a0, a1, a2, a1, a3, a4, a2, a4, a5};
    double U[9];
    svd3_left_vectors(A, U);
    // C = sum_k v_k u_k u_k^T with v = (1, 1, epsilon): the loop of PCL's computeCovariances, in its order
    const int rr[6] = {0, 1, 2, 1, 2, 2}, cc[6] = {0, 0, 0, 1, 1, 2};  // (0,0) (1,0) (2,0) (1,1) (2,1) (2,2)
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double v = (k == 2) ? kGicpEpsilon : 1.0;
        acc += v * U[3 * rr[e] + k] * U[3 * cc[e] + k];
      }
      C[e] = acc;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) c[k] = C[k];
}

// ---- Mahalanobis matrices --------------------------------------------------------------------------------------
// M = (C_t + R C_s R^T)^-1 of one correspondence, upper triangle (a, b: the two covariances' upper triangles)
__device__ __forceinline__ void gicp_maha_of(const double* __restrict__ a, const double* __restrict__ b, const Rot3d& R, double* __restrict__ M) {
  const double C1[9] = {a[0], a[1], a[2], a[1], a[3], a[4], a[2], a[4], a[5]};
  double RC[9], S[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) RC[3 * r + c] = R.m[3 * r] * C1[c] + R.m[3 * r + 1] * C1[3 + c] + R.m[3 * r + 2] * C1[6 + c];
  const double C2[9] = {b[0], b[1], b[2], b[1], b[3], b[4], b[2], b[4], b[5]};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      S[3 * r + c] = RC[3 * r] * R.m[3 * c] + RC[3 * r + 1] * R.m[3 * c + 1] + RC[3 * r + 2] * R.m[3 * c + 2] + C2[3 * r + c];
  // inverse by adjugate; the result of inverting a (numerically) symmetric matrix is stored as its upper triangle
  const double c00 = S[4] * S[8] - S[5] * S[7], c01 = S[5] * S[6] - S[3] * S[8], c02 = S[3] * S[7] - S[4] * S[6];
  const double id = 1.0 / (S[0] * c00 + S[1] * c01 + S[2] * c02);
  M[0] = c00 * id;
  M[1] = (S[2] * S[7] - S[1] * S[8]) * id;
  M[2] = (S[1] * S[5] - S[2] * S[4]) * id;
  M[3] = (S[0] * S[8] - S[2] * S[6]) * id;
  M[4] = (S[2] * S[3] - S[0] * S[5]) * id;
  M[5] = (S[0] * S[4] - S[1] * S[3]) * id;
}

__global__ __launch_bounds__(256) void gicp_maha_kernel(int n_s, const unsigned long long* __restrict__ keys, float thr,
                                                        Rot3d R, const double* __restrict__ cov_s,
                                                        const double* __restrict__ cov_t, double* __restrict__ maha6) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_s) return;
  const unsigned long long key = keys[i];
  const unsigned int j = (unsigned int)key;
  const float d2 = __uint_as_float((unsigned int)(key >> 32));
  if (j == 0xFFFFFFFFu || !(d2 < thr)) return;
  gicp_maha_of(cov_s + (size_t)i * 6, cov_t + (size_t)j * 6, R, maha6 + (size_t)i * 6);
}

// ---- one BFGS evaluation ---------------------------------------------------------------------------------------
// terms: 0 = m, 1 = sum r^T M r, 2..4 = sum M r, 5..13 = sum (base p)(M r)^T (row-major), 14 = sum d2 of the NN sweep
// ---- order-independent sums ----------------------------------------------------------------------------------------
// BFGS is a chaotic consumer: one ulp in a cost or gradient sum can change a line-search decision, and the outer loop's
// delta < 1 stop sits on a 1e-6 m threshold.  With plain float64 sums the GPU (per-lane partial sums, trees) and the
// oracle (one sequential sum, like PCL) agreed to the last bit on most pairs but ended up to 6 cm apart on 2 % of 1 000
// random pairs (scripts/gicp_campaign.py).  So the 13 real sums of an evaluation (f, g_t, R) are accumulated as
// double-double numbers everywhere -- error-free TwoSum per term in the lane, double-double adds across lanes, waves,
// workgroups and on the host; the oracle keeps a three-fold expansion over its sequential loop -- and rounded to
// float64 once at the end.  Either way the result is the correctly rounded exact sum unless that sum lies within
// ~1e-30 (relative to the sum of magnitudes) of a rounding boundary: the order of summation no longer matters.
struct DD {
  double hi, lo;
};
__host__ __device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}
__host__ __device__ __forceinline__ void dd_add_term(DD& a, double t) {  // a += t, error-free in (hi, lo) up to lo's rounding
  double s, e;
  two_sum(a.hi, t, s, e);
  a.hi = s;
  a.lo += e;
}
__host__ __device__ __forceinline__ DD dd_add(const DD& a, const DD& b) {
  // cascaded: the high parts by TwoSum (error-free), everything small in plain float64 -- the low part stays ~1e-16 of the
  // running magnitudes, so its own rounding is ~1e-32 of them (the accurate double-double addition with two
  // renormalisations costs three times as much and buys nothing here)
  DD r;
  double e;
  two_sum(a.hi, b.hi, r.hi, e);
  r.lo = (a.lo + b.lo) + e;
  return r;
}

// Sixteen double-double numbers added as a BALANCED TREE (depth 4) rather than one after the other: the same fifteen additions,
// but four dependent ones instead of fifteen -- a dependent float64 operation costs a wave 8-10 cycles, an independent one 4
// (round 4: a workgroup's two reduction stages 1.55 -> 0.9 us).  The order of these additions does not matter to the result (the
// sums are exact to ~1e-30: see above), which is what makes them comparable bit for bit with the oracle in the first place.
__host__ __device__ __forceinline__ DD dd_sum16(const DD* x) {
  DD a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = dd_add(x[2 * i], x[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = dd_add(a[2 * i], a[2 * i + 1]);
  a[0] = dd_add(a[0], a[1]);
  a[1] = dd_add(a[2], a[3]);
  return dd_add(a[0], a[1]);
}

constexpr int kGicpSums = 13;  // f, g_t(3), R(9): terms 1..13 of the layout above

struct GicpAcc {
  double m, d2;       // count and sum of the sweep's d2 (sums of floats: exact in float64 at these sizes)
  DD s[kGicpSums];
};

// Four correspondences of a lane (positions i0, i0 + stride, ...): everything an evaluation reads from memory.  Their loads
// are issued together: key -> (target point, Mahalanobis matrix) is a dependent chain of random reads, and one after the
// other they made the evaluation 14-16 us long (latency, not bandwidth).
struct GicpQuad {
  bool use[4];
  float d2[4];
  float4 s[4], q[4];
  double M[4][6];
};
__device__ __forceinline__ void gicp_load_quad(GicpQuad& L, int i0, int stride, const float4* __restrict__ src, int n_s,
                                               const float4* __restrict__ tgt, const unsigned long long* __restrict__ keys,
                                               float thr, const double* __restrict__ maha6) {
  unsigned long long key[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u * stride;
    key[u] = i < n_s ? keys[i] : kEmptyKey;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = min(i0 + u * stride, n_s - 1);
    const unsigned int j = (unsigned int)key[u];
    L.d2[u] = __uint_as_float((unsigned int)(key[u] >> 32));
    L.use[u] = j != 0xFFFFFFFFu && L.d2[u] < thr;
    L.s[u] = src[i];
    L.q[u] = tgt[L.use[u] ? j : 0u];
#pragma unroll
    for (int k = 0; k < 6; ++k) L.M[u][k] = maha6[(size_t)i * 6 + k];
  }
}
__device__ __forceinline__ void gicp_add_quad(GicpAcc& acc, const GicpQuad& L, const Xform& T, const Xform& base) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!L.use[u]) continue;
    const float4 s = L.s[u], q = L.q[u];
    float px, py, pz, bx, by, bz;
    xform_point(T, s.x, s.y, s.z, px, py, pz);
    xform_point(base, s.x, s.y, s.z, bx, by, bz);
    const double r0 = (double)(px - q.x), r1 = (double)(py - q.y), r2 = (double)(pz - q.z);
    const double t0 = L.M[u][0] * r0 + L.M[u][1] * r1 + L.M[u][2] * r2;
    const double t1 = L.M[u][1] * r0 + L.M[u][3] * r1 + L.M[u][4] * r2;
    const double t2 = L.M[u][2] * r0 + L.M[u][4] * r1 + L.M[u][5] * r2;
    acc.m += 1.0;
    dd_add_term(acc.s[0], r0 * t0 + r1 * t1 + r2 * t2);
    dd_add_term(acc.s[1], t0);
    dd_add_term(acc.s[2], t1);
    dd_add_term(acc.s[3], t2);
    const double pb[3] = {(double)bx, (double)by, (double)bz};
    const double tt[3] = {t0, t1, t2};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) dd_add_term(acc.s[4 + 3 * r + c], pb[r] * tt[c]);
    acc.d2 += (double)L.d2[u];
  }
}
__device__ __forceinline__ void gicp_clear(GicpAcc& acc) {
  acc.m = acc.d2 = 0.0;
#pragma unroll
  for (int k = 0; k < kGicpSums; ++k) acc.s[k].hi = acc.s[k].lo = 0.0;
}

__device__ __forceinline__ void gicp_accumulate(GicpAcc& acc, const float4* __restrict__ src, int n_s,
                                                const float4* __restrict__ tgt,
                                                const unsigned long long* __restrict__ keys, float thr, const Xform& T,
                                                const Xform& base, const double* __restrict__ maha6, int wid = (int)blockIdx.x,
                                                int nw = (int)gridDim.x) {
  gicp_clear(acc);
  const int stride = nw * 256;
  for (int i0 = wid * 256 + threadIdx.x; i0 < n_s; i0 += 4 * stride) {
    GicpQuad L;
    gicp_load_quad(L, i0, stride, src, n_s, tgt, keys, thr, maha6);
    gicp_add_quad(acc, L, T, base);
  }
}

// Workgroup reduction of the lanes' accumulators into partials[block * kGicpPartialStride + ...] as four 64-byte ANSWER LINES of
// seven values + one tag (word 8 L + 7): value 0 = m, 1..13 = the sums' high parts, 14 = sum d2, 15..27 = their low parts.  The 13 double-double sums go through LDS: thread
// (sum, chunk) adds the 16 lanes of its chunk in lane order, then thread `sum` adds the 16 chunk results in order.
// md (nullable): the workgroup's m and sum d2 from an EARLIER evaluation of the same correspondences (they do not depend on the
// state): md[2] != 0 means md[0], md[1] are valid and the two wave reductions (twelve dependent cross-lane steps) are skipped;
// otherwise they are computed and left there (thread kGicpSums keeps them).
__device__ __forceinline__ void gicp_block_reduce_store(const GicpAcc& acc, double* __restrict__ partials, unsigned long long tag,
                                                        double* md = nullptr) {
  __shared__ DD s_lane[kGicpSums][16][17];  // [sum][lane % 16][lane / 16], rows padded: neither the writes (a lane per
                                            // thread) nor the reads (a chunk per thread) pile up on one LDS bank
  __shared__ DD s_chunk[kGicpSums][16];
  __shared__ double s_md[2][4];
#pragma unroll
  for (int k = 0; k < kGicpSums; ++k) s_lane[k][threadIdx.x & 15][threadIdx.x >> 4] = acc.s[k];
  const bool have_md = md && md[2] != 0.0;  // (workgroup-uniform: every thread holds the same flag)
  if (!have_md) {
    const double m = wave_sum(acc.m), d2 = wave_sum(acc.d2);
    if ((threadIdx.x & 63) == 0) {
      s_md[0][threadIdx.x >> 6] = m;
      s_md[1][threadIdx.x >> 6] = d2;
    }
  }
  __syncthreads();
  if (threadIdx.x < kGicpSums * 16) {
    const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
    DD x[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) x[l] = s_lane[k][l][c];  // lanes 16c .. 16c+15, in lane order; 16 independent reads, then the tree
    const DD v = dd_sum16(x);
    s_chunk[k][c] = v;
  }
  __syncthreads();
  double* out = partials + (size_t)blockIdx.x * kGicpPartialStride;
  if (threadIdx.x < 64) {
    // wave 0: lane s < 13 adds the chunks of sum s, lane 13 has m and sum d2; then the 28 results are dealt out as four ANSWER
    // LINES of 64 bytes (icp_kernels.h: kGicpLine*): lanes 8 L .. 8 L + 6 hold values 7 L .. 7 L + 6, lane 8 L + 7 the line's tag =
    // (evaluation number << 24) | the XOR of the values' 24-bit folds, and ONE 8-byte store instruction of 32 lanes writes the
    // workgroup's whole answer -- four 64-byte requests towards the host where the 16-byte {value, tag} pairs of rounds 2-4
    // took seven (the answers of a run's workgroups reach the host over 1.2-2 us, the largest single item of an evaluation's
    // wall time, and the host's poll looks at 4 tags per workgroup instead of 28).  A line is valid or recognisably not in
    // whatever order its bytes become visible, exactly like a pair.
    DD v{0.0, 0.0};
    double m = 0.0, d2 = 0.0;
    if (threadIdx.x < kGicpSums) {
      DD x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = s_chunk[threadIdx.x][c];
      v = dd_sum16(x);
    } else if (threadIdx.x == kGicpSums) {
      m = have_md ? md[0] : (s_md[0][0] + s_md[0][1]) + (s_md[0][2] + s_md[0][3]);
      d2 = have_md ? md[1] : (s_md[1][0] + s_md[1][1]) + (s_md[1][2] + s_md[1][3]);
      if (md) {
        md[0] = m;
        md[1] = d2;
      }
    }
    const int lane = (int)threadIdx.x, pos = lane & 7;
    const int slot = 7 * (lane >> 3) + pos;  // value number (pos < 7): 0 = m, 1..13 = high parts, 14 = sum d2, 15..27 = low parts
    const double from_hi = __shfl(v.hi, min(max(slot - 1, 0), kGicpSums - 1), 64), from_lo = __shfl(v.lo, min(max(slot - 15, 0), kGicpSums - 1), 64);
    const double from_m = __shfl(m, kGicpSums, 64), from_d2 = __shfl(d2, kGicpSums, 64);
    const double value = slot == 0 ? from_m : slot <= 13 ? from_hi : slot == 14 ? from_d2 : from_lo;
    const unsigned long long bits = pos < 7 ? (unsigned long long)__double_as_longlong(value) : 0ull;
    unsigned int fold = (unsigned int)((bits ^ (bits >> 24) ^ (bits >> 48)) & 0xFFFFFFull);
    fold ^= (unsigned int)__shfl_xor((int)fold, 1, 64);
    fold ^= (unsigned int)__shfl_xor((int)fold, 2, 64);
    fold ^= (unsigned int)__shfl_xor((int)fold, 4, 64);  // every lane of the line: the XOR of its seven folds
    const unsigned long long word = pos < 7 ? bits : (((tag & ~kMailboxReleaseBit) << 24) | (unsigned long long)fold);
    unsigned long long* w = reinterpret_cast<unsigned long long*>(out) + lane;
    // (ASSUMPTION of the classic form, gfx9 hardware rather than the HIP memory model: the values of a line are stored by lanes
    //  8 L .. 8 L + 6 and its tag by lane 8 L + 7 of the SAME wave; a release fence is executed wave-wide on this hardware
    //  (s_waitcnt + write-back), so the tag lane's release also orders the other lanes' stores.  The model only promises that for
    //  the storing lane itself.  The default form below does not depend on it -- a line carries its own checksum -- and
    //  tests/test_gpu_mailbox.py runs both forms against each other.)
    if (tag & kMailboxReleaseBit) {  // the classic form: values, system-scope release, tags
      if (lane < 8 * kGicpLines && pos < 7) __hip_atomic_store(w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __atomic_thread_fence(__ATOMIC_RELEASE);
      if (lane < 8 * kGicpLines && pos == 7) __hip_atomic_store(w, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (lane < 8 * kGicpLines) {
      __hip_atomic_store(w, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // written through to system memory (sc0 sc1)
    }
  }
  if (md) md[2] = 1.0;
}

// One evaluation as ONE kernel of <= kGicpDirectBlocks workgroups: every workgroup stores its partials straight into
// host-mapped memory (host_partials) followed by host_flags[block] = seq; the host adds the workgroups' partials.  All
// the stores above come from wave 0, like the flag, so the release orders them.
__global__ __launch_bounds__(256) void gicp_cost_kernel(const float4* __restrict__ src, int n_s,
                                                        const float4* __restrict__ tgt,
                                                        const unsigned long long* __restrict__ keys, float thr, Xform T,
                                                        Xform base, const double* __restrict__ maha6,
                                                        double* __restrict__ partials, unsigned long long* flags,
                                                        unsigned long long seq) {
  GicpAcc acc;
  gicp_accumulate(acc, src, n_s, tgt, keys, thr, T, base, maha6);
  (void)flags;
  gicp_block_reduce_store(acc, partials, seq);
}

// ---- the quadratic form of an outer iteration (icp_gicp_quadratic.h) ------------------------------------------------------------
// ONE pass over the correspondences per outer iteration instead of one per BFGS evaluation: the 73 coefficient sums of the cost as
// a quadratic form in the entries of T, plus m and sum d2 -- sum number n of kGicpQuadSums:
//   n = pair(e, e') * 6 + tri(c, d)   sum p~_e p~_e' M_cd     (pairs e <= e' of p~ = (p.x, p.y, p.z, 1); M's upper triangle)   0..59
//   n = 60 + e * 3 + c                 sum p~_e (M q)_c                                                                       60..71
//   n = 72                             sum q^T M q                                 n = 73: m (count)        n = 74: sum d2
// The Mahalanobis matrix of a correspondence is computed here (gicp_maha_of: the operations of gicp_maha_kernel) and never stored.
// Every sum is an exact double-double sum of its float64 terms, like the sums of an evaluation.  A thread keeps its first
// correspondence in registers and walks the sums thirteen at a time (the workgroup reduction's LDS tile holds thirteen), so clouds
// up to 256 x gridDim.x points -- the reference's voxel-filtered scans -- read their inputs once.
struct QuadPoint {
  bool use;
  double p[4], q[3], M[6], Mq[3], cq, d2;
};
__device__ __forceinline__ void quad_point_load(QuadPoint& P, int i, const float4* __restrict__ src, const float4* __restrict__ tgt,
                                                const unsigned long long* __restrict__ keys, float thr, const Rot3d& R,
                                                const double* __restrict__ cov_s, const double* __restrict__ cov_t) {
  const unsigned long long key = keys[i];
  const unsigned int j = (unsigned int)key;
  const float d2 = __uint_as_float((unsigned int)(key >> 32));
  P.use = j != 0xFFFFFFFFu && d2 < thr;
  if (!P.use) return;
  const float4 s = src[i], q = tgt[j];
  gicp_maha_of(cov_s + (size_t)i * 6, cov_t + (size_t)j * 6, R, P.M);
  P.p[0] = (double)s.x; P.p[1] = (double)s.y; P.p[2] = (double)s.z; P.p[3] = 1.0;
  P.q[0] = (double)q.x; P.q[1] = (double)q.y; P.q[2] = (double)q.z;
  P.Mq[0] = P.M[0] * P.q[0] + P.M[1] * P.q[1] + P.M[2] * P.q[2];
  P.Mq[1] = P.M[1] * P.q[0] + P.M[3] * P.q[1] + P.M[4] * P.q[2];
  P.Mq[2] = P.M[2] * P.q[0] + P.M[4] * P.q[1] + P.M[5] * P.q[2];
  P.cq = P.q[0] * P.Mq[0] + P.q[1] * P.Mq[1] + P.q[2] * P.Mq[2];
  P.d2 = (double)d2;
}
template <int N>
__device__ __forceinline__ double quad_term(const QuadPoint& P) {  // term N of one correspondence
  if constexpr (N < 60) {
    constexpr int pair = N / 6, k = N % 6;
    constexpr int e = pair < 4 ? 0 : pair < 7 ? 1 : pair < 9 ? 2 : 3;
    constexpr int f = pair < 4 ? pair : pair < 7 ? pair - 3 : pair < 9 ? pair - 5 : 3;
    return (P.p[e] * P.p[f]) * P.M[k];  // the product of two floats is exact
  } else if constexpr (N < 72) {
    return P.p[(N - 60) / 3] * P.Mq[(N - 60) % 3];
  } else if constexpr (N == 72) {
    return P.cq;
  } else if constexpr (N == 73) {
    return 1.0;
  } else {
    return P.d2;
  }
}
template <int G, int K>
__device__ __forceinline__ void quad_set_group(DD (&acc)[13], const QuadPoint& P) {  // acc = the terms of a lane's FIRST correspondence
  if constexpr (K < 13 && 13 * G + K < kGicpQuadSums) {
    acc[K].hi = quad_term<13 * G + K>(P);
    quad_set_group<G, K + 1>(acc, P);
  }
}
template <int G, int K>
__device__ __forceinline__ void quad_add_group(DD (&acc)[13], const QuadPoint& P) {
  if constexpr (K < 13 && 13 * G + K < kGicpQuadSums) {
    dd_add_term(acc[K], quad_term<13 * G + K>(P));
    quad_add_group<G, K + 1>(acc, P);
  }
}
// partials[block][kGicpQuadSums] (DD); the last workgroup to finish adds the workgroups' partials and publishes the kGicpQuadSums
// (hi, lo) pairs as 2 x kGicpQuadSums result pairs numbered seq in host_out (gicp_granule_read on the host)
template <int G>
__device__ __forceinline__ void quad_group(const QuadPoint& mine, const QuadPoint& mine2, int i_next, int stride, const float4* __restrict__ src, int n_s,
                                           const float4* __restrict__ tgt, const unsigned long long* __restrict__ keys, float thr,
                                           const Rot3d& R, const double* __restrict__ cov_s, const double* __restrict__ cov_t,
                                           DD* __restrict__ out, DD (*s_lane)[16][17], DD (*s_chunk)[16]) {
  DD acc[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) acc[k].hi = acc[k].lo = 0.0;
  if (mine.use) quad_set_group<G, 0>(acc, mine);  // (0 + t by TwoSum is t, seven operations later)
  if (mine2.use) quad_add_group<G, 0>(acc, mine2);
  for (int i = i_next; i < n_s; i += stride) {  // larger clouds: the further correspondences of this lane, read again per group
    QuadPoint P;
    quad_point_load(P, i, src, tgt, keys, thr, R, cov_s, cov_t);
    if (P.use) quad_add_group<G, 0>(acc, P);
  }
#pragma unroll
  for (int k = 0; k < 13; ++k) s_lane[k][threadIdx.x & 15][threadIdx.x >> 4] = acc[k];
  __syncthreads();
  if (threadIdx.x < 13 * 16) {
    const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
    DD x[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) x[l] = s_lane[k][l][c];
    s_chunk[k][c] = dd_sum16(x);
  }
  __syncthreads();
  if (threadIdx.x < 13 && 13 * G + (int)threadIdx.x < kGicpQuadSums) {
    DD x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) x[c] = s_chunk[threadIdx.x][c];
    out[13 * G + threadIdx.x] = dd_sum16(x);
  }
  // (the next group's writes to s_lane come behind this group's second barrier, its writes to s_chunk behind its own first one)
}
__global__ __launch_bounds__(256) void gicp_quadratic_kernel(const float4* __restrict__ src, int n_s, const float4* __restrict__ tgt,
                                                             const unsigned long long* __restrict__ keys, float thr, Rot3d R,
                                                             const double* __restrict__ cov_s, const double* __restrict__ cov_t,
                                                             DD* __restrict__ partials, unsigned int* __restrict__ done,
                                                             unsigned long long* __restrict__ host_out, unsigned long long seq, int stamps) {
  // stamps (development flavour, ICPGPU_GICP_TIMING): the LAST workgroup's 100 MHz clock at six points, as result pairs behind the sums
  const long long st0 = stamps ? (long long)wall_clock64() : 0;
  __shared__ DD s_lane[13][16][17];
  __shared__ DD s_chunk[13][16];
  __shared__ bool s_last;
  const int stride = (int)gridDim.x * 256, i0 = (int)blockIdx.x * 256 + (int)threadIdx.x;
  QuadPoint mine, mine2;  // the lane's first two correspondences stay in registers (half as many workgroups to add up at the end)
  mine.use = mine2.use = false;
  if (i0 < n_s) quad_point_load(mine, i0, src, tgt, keys, thr, R, cov_s, cov_t);
  if (i0 + stride < n_s) quad_point_load(mine2, i0 + stride, src, tgt, keys, thr, R, cov_s, cov_t);
  DD* out = partials + (size_t)blockIdx.x * kGicpQuadSums;
  const long long st1 = stamps ? (long long)wall_clock64() : 0;
  quad_group<0>(mine, mine2, i0 + 2 * stride, stride, src, n_s, tgt, keys, thr, R, cov_s, cov_t, out, s_lane, s_chunk);
  quad_group<1>(mine, mine2, i0 + 2 * stride, stride, src, n_s, tgt, keys, thr, R, cov_s, cov_t, out, s_lane, s_chunk);
  quad_group<2>(mine, mine2, i0 + 2 * stride, stride, src, n_s, tgt, keys, thr, R, cov_s, cov_t, out, s_lane, s_chunk);
  quad_group<3>(mine, mine2, i0 + 2 * stride, stride, src, n_s, tgt, keys, thr, R, cov_s, cov_t, out, s_lane, s_chunk);
  quad_group<4>(mine, mine2, i0 + 2 * stride, stride, src, n_s, tgt, keys, thr, R, cov_s, cov_t, out, s_lane, s_chunk);
  quad_group<5>(mine, mine2, i0 + 2 * stride, stride, src, n_s, tgt, keys, thr, R, cov_s, cov_t, out, s_lane, s_chunk);
  // the last workgroup to get here adds everybody's partials (device-scope release / acquire around the counter)
  const long long st2 = stamps ? (long long)wall_clock64() : 0;
  // (every global store of this workgroup came from threads 0..12 -- wave 0, the wave of thread 0 -- so the counter's RELEASE orders
  //  them by itself: one wave's write-back instead of a fence in all four and a barrier, 3.3 -> ~1.5 us on the last workgroup's path)
  //  Since round 6 the storing lanes say so themselves: an agent-scope release fence in wave 0 (all of its lanes: the fence is
  //  wave-wide on this hardware anyway, and the memory model orders a lane's stores only behind that lane's own release), then
  //  the counter -- ADVICE r5; one more write-back of an already clean wave.)
  if (threadIdx.x < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();  // (no instruction: the lanes of a wave are in step; it marks the wavefront-scope hand-over to lane 0)
  }
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave of the last workgroup: its loads below must not be served from stale lines
  const long long st3 = stamps ? (long long)wall_clock64() : 0;
  if (threadIdx.x == 0) *done = 0u;  // ready for the next launch (launches of one context are ordered by its stream)
  // three threads per sum, each over a third of the workgroups, sixteen loads in flight (one load after the other -- 89 workgroups
  // for a voxel-filtered scan at first -- made this tail ~60 us long: every load a trip to L2 and back before the next one left)
  __shared__ DD s_fin[3][kGicpQuadSums];
  const int slice = (int)threadIdx.x / kGicpQuadSums, k = (int)threadIdx.x % kGicpQuadSums, nb = (int)gridDim.x;
  if (slice < 3) {
    DD a{0.0, 0.0};
    for (int b0 = slice; b0 < nb; b0 += 3 * 16) {  // sixteen loads in flight: a voxel-filtered scan's 45 workgroups in ONE trip
      DD v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int bb = b0 + 3 * u;
        v[u] = bb < nb ? partials[(size_t)bb * kGicpQuadSums + k] : DD{0.0, 0.0};
      }
      a = dd_add(a, dd_sum16(v));
    }
    s_fin[slice][k] = a;
  }
  __syncthreads();
  const long long st4 = stamps ? (long long)wall_clock64() : 0;
  if (stamps && threadIdx.x == 255) {
    const long long t[5] = {st0, st1, st2, st3, st4};
    for (int u = 0; u < 5; ++u) store_result_pair(host_out + 2 * (2 * kGicpQuadSums + u), (unsigned long long)t[u], seq);
  }
  if (threadIdx.x < kGicpQuadSums) {
    const DD a = dd_add(dd_add(s_fin[0][k], s_fin[1][k]), s_fin[2][k]);
    store_result_pair(host_out + 2 * (2 * threadIdx.x), (unsigned long long)__double_as_longlong(a.hi), seq);
    store_result_pair(host_out + 2 * (2 * threadIdx.x + 1), (unsigned long long)__double_as_longlong(a.lo), seq);
  }
}

// ---- resident evaluation server ------------------------------------------------------------------------------------
// A BFGS run is ~35 DEPENDENT evaluations of a few microseconds of work each; launched one by one they cost 11 us apiece,
// most of it launch and dispatch latency.  This kernel stays resident for the whole run instead (<= kGicpDirectBlocks
// workgroups, all co-resident by construction).  The command line -- 12 floats of T, then a sequence number -- lives in
// FINE-GRAINED DEVICE memory that the host writes through the PCIe BAR (posted writes, delivered in order: the number is
// written last, behind a store fence), so every workgroup polls it locally: a poll is an uncached read of HBM, not a PCIe
// round trip (polling a line in HOST memory from one workgroup and relaying it through device memory with release /
// acquire fences was measured first: 13 us per evaluation, slower than a launch).  Every workgroup then evaluates its
// share and stores its 17 partials + flag into the host mailbox exactly like gicp_cost_kernel's direct mode.  Sequence
// number kGicpServerExit ends the run; so does 50 ms without a command (the host fell asleep or died: never leave a
// spinning kernel behind -- the host notices the idle stream and goes back to single launches).
template <bool RESIDENT>
__global__ __launch_bounds__(256) void gicp_server_kernel(const float4* __restrict__ src, int n_s,
                                                          const float4* __restrict__ tgt,
                                                          const unsigned long long* __restrict__ keys, float thr, Xform base,
                                                          const double* __restrict__ maha6, double* __restrict__ host_partials,
                                                          unsigned long long* host_flags, unsigned int* cmd,
                                                          unsigned int first_seq, unsigned int seq_hi, int stamps) {
  __shared__ Xform s_T;
  __shared__ unsigned int s_seq;
  __shared__ long long s_seen;
  __shared__ unsigned int s_polls;
  const long long patience = 5000000;  // 50 ms of the 100 MHz wall clock
  unsigned int expect = first_seq;
  // RESIDENT (small problems: every lane's share is one quad): the correspondences do not change during a run, so they are
  // read ONCE and stay in registers -- an evaluation then starts without the key -> (target point, matrix) chain of reads.
  // (298 registers, one wave per SIMD.  Measured twice: slower at 200k points while every evaluation also paid a system-scope
  // release, 9-11 % faster at every size from 5k to 200k once the results travelled as self-tagged pairs.)
  GicpQuad mine;
  if constexpr (RESIDENT) gicp_load_quad(mine, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, src, n_s, tgt, keys, thr, maha6);
  double md[3] = {0.0, 0.0, 0.0};  // this workgroup's m and sum d2: the same at every evaluation of the run
  for (;;) {
    const long long t_loop = (long long)wall_clock64();
    if (threadIdx.x < 64) {
      // lane k < 13 reads word k of the line, all in ONE coalesced read per poll (the line is uncached: every load reads
      // memory).  The words of T were posted before the number, so a read that returns the number returns them too.
      unsigned int got = kGicpServerExit, word = 0u;
      const long long t0 = (long long)wall_clock64();
      // (One read in flight, 0.18 us each.  Four -- a hand-written loop behind s_waitcnt vmcnt(3) -- were measured in round 4 and
      // made an evaluation SLOWER, 8.3 instead of 7.1 us.  Round 5, the other direction: an s_sleep of 128 / 384 clocks between
      // polls changes nothing either, alone or with eight servers resident: profiles/r05_gicp_batch.txt.)
      const unsigned int* line = &cmd[threadIdx.x & 15];
      unsigned int polls_done = 0;
      for (unsigned int polls = 1;; ++polls) {
        polls_done = polls;
        word = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned int v = (unsigned int)__builtin_amdgcn_readlane((int)word, 12);
        if (v == expect || v == kGicpServerExit) {
          got = v;
          break;
        }
        if ((polls & 15u) == 0 && (long long)wall_clock64() - t0 > patience) break;  // got stays kGicpServerExit
      }
      if (threadIdx.x < 12) s_T.m[threadIdx.x] = __uint_as_float(word);
      if (threadIdx.x == 12) {
        s_seq = got;
        s_seen = (long long)wall_clock64();
        s_polls = polls_done;
      }
    }
    __syncthreads();
    const unsigned int seq = s_seq;
    if (seq == kGicpServerExit) {
      // acknowledge: the host does not reuse the command line before this kernel has left
      if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(&host_flags[0], ~0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    const Xform T = s_T;
    GicpAcc acc;
    if constexpr (RESIDENT) {
      gicp_clear(acc);
      gicp_add_quad(acc, mine, T, base);
    } else {
      gicp_accumulate(acc, src, n_s, tgt, keys, thr, T, base, maha6);
    }
    const long long t_acc = stamps ? (long long)wall_clock64() : 0;
    gicp_block_reduce_store(acc, host_partials, ((unsigned long long)seq_hi << 32) | seq, md);
    if (stamps && threadIdx.x == 0) {  // (development flavour, ICPGPU_GICP_TIMING: three spare entries of this workgroup's line; 100 MHz ticks)
      double* o = host_partials + (size_t)blockIdx.x * kGicpPartialStride;
      o[2 * 29] = (double)s_seen;                    // command seen
      o[2 * 30] = (double)t_acc;                     // its share accumulated
      o[2 * 31] = (double)(long long)wall_clock64(); // reduced, results on their way
      o[2 * 29 + 1] = (double)t_loop;                // started polling
      o[2 * 30 + 1] = (double)s_polls;               // reads of the command line it took
      {
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[2 * 31 + 1] = (double)(xcc & 15u);         // the XCD it runs on
      }
    }
    expect = seq + 1u;
    if (expect == kGicpServerExit) expect = 0u;
    __syncthreads();  // s_T / s_seq are rewritten in the next round
  }
}


// ---- the device solver (round 4) --------------------------------------------------------------------------------------------
// What the reference runs per scan (icp_odometer.cpp:188-198) is ~5 outer iterations of ~35 DEPENDENT cost evaluations each; with
// the BFGS on the host every one of them was a host <-> device round trip (7.6 us wall for 1.4 us of device work: 1.3 of the
// pipeline's 2.2 ms per scan).  gicp_solve_kernel runs the WHOLE inner minimisation of an outer iteration on the device: the
// host queues it behind the search and the Mahalanobis kernels and polls ONE result.
//
//   * Every workgroup holds its share of the correspondences in registers (RESIDENT; streamed from memory when the share is
//     larger than one quad per lane) and runs the SAME solver redundantly -- icp_gicp_solver_impl.h, the source the host
//     fallback instantiates: same operations, same order, correctly rounded sines and cosines (icp_trig.h), so every lane of
//     every workgroup takes the same decisions and the bits equal the host path's and the oracle's.
//   * One evaluation = one hop: each workgroup reduces its share (double-double, two LDS stages), PUBLISHES 28 granules --
//     {value, tag}, one 16-byte write-through store each, tag = evaluation number and a checksum of the value's bits, so a
//     granule is valid or recognisably not, whatever the order its halves become visible in -- into its slot of a fine-grained
//     device buffer, and wave 0 of EVERY workgroup GATHERS all slots (4 lanes per sum, every load of a round in flight at
//     once), merges them in double-double (order-independent to ~1e-30) and hands the 15 numbers to the workgroup through LDS.
//     There is no master and no command hop: the next state is computed everywhere at once.
//   * Slots are double-buffered by the evaluation number's parity: a workgroup can run at most one evaluation ahead of the
//     slowest one (it needs everybody's partials of evaluation k to start k + 1).
//   * Nobody waits for ever: 50 ms without a granule ends the run with kDeviceError everywhere (the host falls back to its own
//     solver); the workgroups are co-resident by construction (one per CU, at most the context's share of the chip).
constexpr int kSolveLocalBlocks = 32;  // CUs of one XCD: the most workgroups of the one-XCD variant
constexpr int kSolveGranules = 28;  // per workgroup and parity: 13 sums' high parts, their low parts, m, sum d2
constexpr int kSolveOut = 20;       // host granules: status, x[6], m, sum d2, f, evaluations, inner iterations done

__device__ __forceinline__ unsigned long long granule_tag(unsigned long long seq, unsigned long long bits) { return mailbox_tag(seq, bits); }
__device__ __forceinline__ void granule_store(unsigned long long* g, double value, unsigned long long seq) {
  store_result_pair(g, (unsigned long long)__double_as_longlong(value), seq);   // (host-visible results: seq may carry the release bit)
}
struct GicpSolveArgs {
  const float4* src;
  int n_s;
  const float4* tgt;
  const unsigned long long* keys;
  float thr;
  Xform base;
  float guess[16];
  const double* maha6;
  double x0[6];
  unsigned long long* slots;     // [2][gridDim.x][kSolveGranules] granules (2 words each), fine-grained device memory
  unsigned long long* host_out;  // kSolveOut granules in the host mailbox
  unsigned long long seq0;       // evaluation e of this run carries the number seq0 + e; the result carries seq0
  unsigned long long host_seq;   // seq0, with kMailboxReleaseBit if the host mailbox is in release mode
  int max_inner;
  double gradient_tol;
  // one-XCD variant (LOCAL): `workers` participants, all on XCD `xcc_want`, exchange their granules through that XCD's L2
  int workers;
  int xcc_want;
  unsigned long long* owner;     // [kGicpDirectBlocks]: worker index -> number of the run that claimed it
};

// The six sine / cosine pairs of a state, six lanes at a time: lane l (mod 8) evaluates argument l, the results come back through
// v_readlane (every group of eight lanes computes the same six, so lanes 0..5 of the wave serve everybody).  Same function, same
// bits as the host's one-after-the-other loop (gicp::trig6), a sixth of the instructions on the critical path.
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane: a compile-time constant
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ gicp::Trig6 trig6_lanes(const gicp::V6& x, const double (*table)[4]) {
  double a[6];
  gicp::trig6_arguments(x, a);
  const int l = (int)threadIdx.x & 7;
  double arg = a[0];
  arg = l == 1 ? a[1] : arg;
  arg = l == 2 ? a[2] : arg;
  arg = l == 3 ? a[3] : arg;
  arg = l == 4 ? a[4] : arg;
  arg = l >= 5 ? a[5] : arg;
  double sv, cv;
  trig::sincos_cr_with(table, arg, &sv, &cv);
  double s6[6], c6[6];
  s6[0] = readlane_f64(sv, 0); c6[0] = readlane_f64(cv, 0);
  s6[1] = readlane_f64(sv, 1); c6[1] = readlane_f64(cv, 1);
  s6[2] = readlane_f64(sv, 2); c6[2] = readlane_f64(cv, 2);
  s6[3] = readlane_f64(sv, 3); c6[3] = readlane_f64(cv, 3);
  s6[4] = readlane_f64(sv, 4); c6[4] = readlane_f64(cv, 4);
  s6[5] = readlane_f64(sv, 5); c6[5] = readlane_f64(cv, 5);
  gicp::Trig6 t;
  gicp::trig6_from(s6, c6, t);
  return t;
}

// LOCAL: the workgroups of the run sit on ONE XCD (each has checked its own XCC id), so a granule travels through the L2 they
// share -- a plain 16-byte store (stays in L2) and L1-bypassing loads (sc1, served by L2): ~0.3 us per hop instead of the
// ~1 us of a trip through the fabric to fine-grained memory and back.  Not LOCAL: any placement, write-through stores and
// system-scope loads on fine-grained memory.  The protocol (self-validating granules, double buffer, timeouts) is the same.
template <bool LOCAL>
__device__ __forceinline__ void granule_publish(unsigned long long* g, double value, unsigned long long seq) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(value);
  if constexpr (LOCAL) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u64x2*>(g) = u64x2{bits, granule_tag(seq, bits)};
  } else {
    store_pair_system(g, bits, granule_tag(seq, bits));
  }
}
template <bool LOCAL>
__device__ __forceinline__ unsigned long long granule_word(const unsigned long long* p) {
  if constexpr (LOCAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#if defined(ICPGPU_DEV_SWITCHES)
#define SOLVE_CLOCK() ((long long)wall_clock64())   // development flavour: phase times of an evaluation (six clock reads of ~0.15 us each)
#else
#define SOLVE_CLOCK() (0ll)
#endif

template <bool RESIDENT, bool LOCAL>
struct DeviceEval {
  const GicpSolveArgs& A;
  const GicpQuad& mine;
  int wid, nw;         // this workgroup's index among the run's workgroups, and their number
  double* s_sums;      // LDS: the 15 numbers of an evaluation
  int* s_ok;           // LDS: the gather's verdict
  const double (*s_table)[4];  // LDS: icp_trig.h's table
  unsigned long long n_eval = 0;
  double m = 0.0, d2 = 0.0;
  double dbg_raw[4] = {0, 0, 0, 0};
  long long t_apply = 0, t_acc = 0, t_pub = 0, t_gather = 0, t_grad = 0;  // phase times (100 MHz ticks), development
  unsigned long long n_pass = 0;  // gather passes (development)
  double md[3] = {0.0, 0.0, 0.0};  // this workgroup's m and sum d2 (the same at every evaluation of the run)
  double dbg = 0.0;    // on a gather timeout: evaluation * 1e6 + the first missing workgroup * 1e3 + quantity (host_out granule 11)

  __device__ __forceinline__ bool operator()(const gicp::V6& x, gicp::Eval& out) {
    const long long c0 = SOLVE_CLOCK();
    const gicp::Trig6 tr = trig6_lanes(x, s_table);
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = A.guess[i];
    gicp::apply_state(T, x, tr);
    Xform Tx;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Tx.m[4 * r + c] = T[c * 4 + r];
    const long long c1 = SOLVE_CLOCK();
    GicpAcc acc;
    if constexpr (RESIDENT) {
      gicp_clear(acc);
      gicp_add_quad(acc, mine, Tx, A.base);
    } else {
      gicp_accumulate(acc, A.src, A.n_s, A.tgt, A.keys, A.thr, Tx, A.base, A.maha6, wid, nw);
    }
    const long long c2 = SOLVE_CLOCK();
    ++n_eval;
    const unsigned long long seq = A.seq0 + n_eval;
    const int B = nw;
    unsigned long long* parity = A.slots + (size_t)(n_eval & 1ull) * (size_t)B * kSolveGranules * 2;
    publish(acc, parity + (size_t)wid * kSolveGranules * 2, seq);
    const long long c3 = SOLVE_CLOCK();
    gather(parity, B, seq);
    __syncthreads();
    const long long c4 = SOLVE_CLOCK();
    const bool ok = *s_ok != 0;
    double s[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) s[k] = s_sums[k];
    __syncthreads();  // (the LDS words are rewritten by the next evaluation)
    if (!ok) return false;
    m = s[0];
    d2 = s[14];
    gicp::eval_from_sums(tr, s, out);
    const long long c5 = SOLVE_CLOCK();
    t_apply += c1 - c0;
    t_acc += c2 - c1;
    t_pub += c3 - c2;
    t_gather += c4 - c3;
    t_grad += c5 - c4;
    return true;
  }

  // workgroup reduction of the lanes' accumulators (the two LDS stages of gicp_block_reduce_store), then 28 granules
  __device__ __forceinline__ void publish(const GicpAcc& acc, unsigned long long* slot, unsigned long long seq) {
    __shared__ DD s_lane[kGicpSums][16][17];
    __shared__ DD s_chunk[kGicpSums][16];
    __shared__ double s_md[2][4];
#pragma unroll
    for (int k = 0; k < kGicpSums; ++k) s_lane[k][threadIdx.x & 15][threadIdx.x >> 4] = acc.s[k];
    const bool have_md = md[2] != 0.0;
    if (!have_md) {
      const double wm = wave_sum(acc.m), wd = wave_sum(acc.d2);
      if ((threadIdx.x & 63) == 0) {
        s_md[0][threadIdx.x >> 6] = wm;
        s_md[1][threadIdx.x >> 6] = wd;
      }
    }
    __syncthreads();
    if (threadIdx.x < kGicpSums * 16) {
      const int k = threadIdx.x >> 4, c = threadIdx.x & 15;
      DD x[16];
#pragma unroll
      for (int l = 0; l < 16; ++l) x[l] = s_lane[k][l][c];
      const DD v = dd_sum16(x);
      s_chunk[k][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < kGicpSums) {
      DD x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) x[c] = s_chunk[threadIdx.x][c];
      const DD v = dd_sum16(x);
      granule_publish<LOCAL>(slot + 2 * threadIdx.x, v.hi, seq);
      granule_publish<LOCAL>(slot + 2 * (kGicpSums + threadIdx.x), v.lo, seq);
    } else if (threadIdx.x == kGicpSums) {
      if (!have_md) {
        md[0] = (s_md[0][0] + s_md[0][1]) + (s_md[0][2] + s_md[0][3]);
        md[1] = (s_md[1][0] + s_md[1][1]) + (s_md[1][2] + s_md[1][3]);
      }
      granule_publish<LOCAL>(slot + 2 * 26, md[0], seq);
      granule_publish<LOCAL>(slot + 2 * 27, md[1], seq);
    }
    md[2] = 1.0;
  }

  // wave 0: lane = (quantity q = lane / 4, part p = lane % 4) merges the workgroups p, p + 4, ... of quantity q, then the four
  // parts are combined by two exchanges; quantities 0..12 are the double-double sums, 13 = m, 14 = sum d2 (plain sums of
  // integers / of floats: exact in float64 at these sizes).  s_sums[0] = m, [1..13] = the sums rounded once, [14] = sum d2.
  __device__ __forceinline__ void gather(const unsigned long long* parity, int B, unsigned long long seq) {
    if (threadIdx.x >= 64) return;
    const int q = (int)threadIdx.x >> 2, p = (int)threadIdx.x & 3;
    const bool active = q < 15;
    const int g_hi = q < kGicpSums ? q : (q == 13 ? 26 : 27);
    const bool two = q < kGicpSums;
    const int g_lo = two ? kGicpSums + q : g_hi;  // (quantities without a low part read their one granule twice)
    DD v{0.0, 0.0};
    bool failed = false;
    const long long patience = 5000000;  // 50 ms of the 100 MHz wall clock
    constexpr int R = 8;                 // workgroups per lane and round: 2 R loads in flight
    for (int b0 = p; b0 < B && !failed; b0 += 4 * R) {
      double hi[R], lo[R];
      unsigned int have = 0u, want = 0u;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        hi[u] = lo[u] = 0.0;
        if (active && b0 + 4 * u < B) want |= 1u << u;
      }
      long long t0 = 0;
      for (unsigned pass = 0; have != want; ++pass) {
        ++n_pass;
        // every load of the round is issued before any is looked at (addresses clamped to the last workgroup: no branches
        // between the loads), then the granules are validated; what is not there yet is read again
        unsigned long long hb[R], ht[R], lb[R], lt[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
          hb[u] = ht[u] = lb[u] = lt[u] = 0ull;
          if (b0 - p + 4 * u >= B) continue;  // (wave-uniform: nobody's u-th workgroup of this round exists)
          const int b = min(b0 + 4 * u, B - 1);
          const unsigned long long* slot = parity + (size_t)b * kSolveGranules * 2;
          hb[u] = granule_word<LOCAL>(slot + 2 * g_hi);
          ht[u] = granule_word<LOCAL>(slot + 2 * g_hi + 1);
          lb[u] = granule_word<LOCAL>(slot + 2 * g_lo);
          lt[u] = granule_word<LOCAL>(slot + 2 * g_lo + 1);
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const bool ok = ht[u] == granule_tag(seq, hb[u]) && (!two || lt[u] == granule_tag(seq, lb[u]));
          if (ok && (((want & ~have) >> u) & 1u)) {
            hi[u] = __longlong_as_double((long long)hb[u]);
            lo[u] = two ? __longlong_as_double((long long)lb[u]) : 0.0;
            have |= 1u << u;
          }
        }
        if (have != want && (pass & 31u) == 31u) {  // (the clock is a memory instruction: not in every pass)
          const long long now = (long long)wall_clock64();
          if (t0 == 0) t0 = now;
          if (now - t0 <= patience) continue;
          failed = true;
          int miss = 0;
#pragma unroll
          for (int u = R - 1; u >= 0; --u)
            if (((want & ~have) >> u) & 1u) miss = b0 + 4 * u;
          dbg = (double)n_eval * 1e6 + (double)miss * 1e3 + (double)q;
          break;
        }
      }
      {  // the round's (up to) eight workgroups as a tree (absent ones are zeros), then onto the running sum
        DD t[R];
#pragma unroll
        for (int u = 0; u < R; ++u) t[u] = ((want >> u) & 1u) ? DD{hi[u], lo[u]} : DD{0.0, 0.0};
#pragma unroll
        for (int u = 0; u < R / 2; ++u) t[u] = dd_add(t[2 * u], t[2 * u + 1]);
#pragma unroll
        for (int u = 0; u < R / 4; ++u) t[u] = dd_add(t[2 * u], t[2 * u + 1]);
        v = dd_add(v, dd_add(t[0], t[1]));
      }
    }
    // combine the four parts (TwoSum is symmetric in its arguments: all four lanes end with the same bits)
#pragma unroll
    for (int d = 1; d <= 2; d <<= 1) {
      DD o;
      o.hi = __shfl_xor(v.hi, d, 64);
      o.lo = __shfl_xor(v.lo, d, 64);
      v = dd_add(v, o);
    }
    const bool any_failed = __any(failed ? 1 : 0) != 0;
    if (any_failed) {  // the lowest failing lane's record to lane 0
      const unsigned long long mask = __ballot(failed ? 1 : 0);
      const int src_lane = __ffsll((long long)mask) - 1;
      dbg = __shfl(dbg, src_lane, 64);
      for (int k = 0; k < 4; ++k) dbg_raw[k] = __shfl(dbg_raw[k], src_lane, 64);
    }
    if (active && p == 0) s_sums[q < kGicpSums ? 1 + q : (q == 13 ? 0 : 14)] = v.hi + v.lo;
    if (threadIdx.x == 0) *s_ok = any_failed ? 0 : 1;
  }
};

// The body of a run: `bx` of `n_launched` workgroups (blockIdx.x / gridDim.x of the single-run launch; in a batched launch the
// run's own numbers).
template <bool RESIDENT, bool LOCAL>
__device__ __forceinline__ void gicp_solve_body(const GicpSolveArgs& A, int bx, int n_launched) {
  __shared__ double s_sums[16];
  __shared__ int s_ok;
  __shared__ double s_table[128][4];
  // (Measured and dropped in round 5: s_setprio 3 for this latency-bound wave and the evaluation server's -- no effect on an
  //  evaluation's time, alone or with eight registrations in flight: profiles/r05_gicp_batch.txt.)
  const long long t_kernel0 = (long long)wall_clock64();
  int wid = bx, nw = n_launched;
  if constexpr (LOCAL) {
    // Eight times the workgroups the run needs are launched; a workgroup takes part iff the hardware says it sits on XCD
    // xcc_want -- round-robin dispatch puts workgroup b on XCD b % 8, so those are the ones with one residue and b / 8 numbers
    // them 0 .. workers - 1, but NOTHING relies on that: a worker index is claimed (atomic exchange of the run's number), a
    // second claimant leaves, and an index nobody claims makes the gathers time out -- the host then uses the variant that
    // works under any placement.
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 0xFu) != A.xcc_want) return;
    wid = bx >> 3;
    nw = A.workers;
    if (wid >= nw) return;
    __shared__ int s_mine;
    if (threadIdx.x == 0) s_mine = atomicExch(&A.owner[wid], A.seq0) != A.seq0 ? 1 : 0;
    __syncthreads();
    if (!s_mine) return;
  }
  {
    const double (*src_table)[4] = trig::trig_table();
    for (int i = threadIdx.x; i < 512; i += 256) s_table[i >> 2][i & 3] = src_table[i >> 2][i & 3];
    __syncthreads();
  }
  GicpQuad mine;
  if constexpr (RESIDENT)
    gicp_load_quad(mine, wid * 256 + threadIdx.x, nw * 256, A.src, A.n_s, A.tgt, A.keys, A.thr, A.maha6);
  DeviceEval<RESIDENT, LOCAL> ev{A, mine, wid, nw, s_sums, &s_ok, s_table};
  gicp::V6 x;
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = A.x0[i];
  gicp::Eval probe;
  int status;
  double f_last = 0.0;
  if (!ev(x, probe)) {
    status = gicp::kDeviceError;
  } else if (!(ev.m >= 4.0)) {
    status = gicp::kNotEnoughPoints;  // PCL: NotEnoughPointsException (fewer than 4 correspondences)
  } else {
    const double m0 = ev.m, d20 = ev.d2;
    status = (int)gicp::minimize(ev, x, A.max_inner, A.gradient_tol, &probe);
    ev.m = m0;  // (the correspondences do not change during a run: every evaluation counts the same m and sum d2)
    ev.d2 = d20;
  }
  f_last = probe.f;
  if (wid == 0 && threadIdx.x == 0) {
    unsigned long long* o = A.host_out;
    granule_store(o + 2 * 1, x[0], A.host_seq);
    granule_store(o + 2 * 2, x[1], A.host_seq);
    granule_store(o + 2 * 3, x[2], A.host_seq);
    granule_store(o + 2 * 4, x[3], A.host_seq);
    granule_store(o + 2 * 5, x[4], A.host_seq);
    granule_store(o + 2 * 6, x[5], A.host_seq);
    granule_store(o + 2 * 7, ev.m, A.host_seq);
    granule_store(o + 2 * 8, ev.d2, A.host_seq);
    granule_store(o + 2 * 9, f_last, A.host_seq);
    granule_store(o + 2 * 10, (double)ev.n_eval, A.host_seq);
    granule_store(o + 2 * 11, ev.dbg, A.host_seq);
    if (ev.dbg != 0.0) {
      for (int k = 0; k < 4; ++k) granule_store(o + 2 * (12 + k), ev.dbg_raw[k], A.host_seq);
      for (int k = 16; k < 18; ++k) granule_store(o + 2 * k, 0.0, A.host_seq);
    } else {  // development: phase times in microseconds, packed two to a granule (apply | accumulate, publish | gather, gradient | total)
      const long long t_all = (long long)wall_clock64() - t_kernel0;
      granule_store(o + 2 * 12, (double)ev.t_apply * 0.01, A.host_seq);
      granule_store(o + 2 * 13, (double)ev.t_acc * 0.01, A.host_seq);
      granule_store(o + 2 * 14, (double)ev.t_pub * 0.01, A.host_seq);
      granule_store(o + 2 * 15, (double)ev.t_gather * 0.01, A.host_seq);
      granule_store(o + 2 * 16, (double)ev.t_grad * 0.01, A.host_seq);
      granule_store(o + 2 * 17, (double)t_all * 0.01, A.host_seq);
    }
    granule_store(o + 2 * 18, (double)ev.n_pass, A.host_seq);
    granule_store(o + 2 * 19, 0.0, A.host_seq);
    granule_store(o + 2 * 0, (double)status, A.host_seq);
  }
}

template <bool RESIDENT, bool LOCAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gicp_solve_kernel(GicpSolveArgs A) {
  gicp_solve_body<RESIDENT, LOCAL>(A, (int)blockIdx.x, (int)gridDim.x);
}

// Several runs in ONE launch (icpgpu_align_batch in GICP mode): run = blockIdx.y, its own workgroups 0 .. workers - 1 (the
// others leave at once), its own slots, mailbox and numbers -- the runs share nothing but the launch.  Why: the runtime puts
// streams on four hardware queues and kernels that share a queue run one after the other, so with one resident solver kernel
// per run and outer iteration no more than four registrations ever solved at the same time (1.7k pairs/s whatever the
// scheduler, round 5: profiles/r05_gicp_batch.txt); one launch for all the runs that are ready lifts that.  Any placement
// (fine-grained slots): the one-XCD variant's placement trick does not compose with blockIdx.y.
struct GicpSolveBatchArgs {
  GicpSolveArgs a[kGicpSolveBatchMax];
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gicp_solve_batch_kernel(GicpSolveBatchArgs B) {
  const GicpSolveArgs& A = B.a[blockIdx.y];
  if ((int)blockIdx.x >= A.workers) return;
  gicp_solve_body<true, false>(A, (int)blockIdx.x, A.workers);
}

}  // namespace

// list (optional): 2 n + 2 ints of scratch -- with it the selecting kernel runs first, the far-field kernel takes what that one hands
// over (clouds up to kGicpCovFarMost points; in larger ones gicp_cov_kernel finishes the list); without it gicp_cov_kernel does
// every point, as until round 5.
hipError_t launch_gicp_covariances(const float4* cloud, int n, const float4* sorted, const int* cell_start,
                                   const GridDesc& g, double* cov6, hipStream_t stream, int* list, bool list_counters_zero) {
  if (n <= 0) return hipSuccess;
  // development flavour, ICPGPU_COV_SELECT=0: the streaming kernel for every point
  static const bool select = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_COV_SELECT"); return !e || atoi(e) != 0; }();
  if (list && select) {
    if (!list_counters_zero) {  // (a fresh buffer; afterwards the finish kernel leaves them zero: one launch less per cloud)
      hipError_t e = hipMemsetAsync(list, 0, 2 * sizeof(int), stream);
      if (e != hipSuccess) return e;
    }
    // development flavour, ICPGPU_COV_STATS=1: what the selecting kernel did with the cloud (a synchronising print per cloud)
    static const bool want_stats = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_COV_STATS"); return e && atoi(e) != 0; }();
    unsigned long long* d_stats = nullptr;
    if (want_stats && hipMalloc(reinterpret_cast<void**>(&d_stats), 16 * sizeof(unsigned long long)) == hipSuccess)
      (void)hipMemsetAsync(d_stats, 0, 16 * sizeof(unsigned long long), stream);
    // list layout: [0] streaming count, [1] far count, [2 .. 2 + n) streaming list, [2 + n .. 2 + 2n) far list
    const int far_ok = n <= kGicpCovFarMost ? 1 : 0;
    int* far_list = list + 2 + n;
    if (d_stats)
      hipLaunchKernelGGL(gicp_cov_select_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, stream, cloud, n, sorted, cell_start, g, cov6, list + 2, list, far_ok,
                         far_list, list + 1, d_stats);
    else
      hipLaunchKernelGGL(gicp_cov_select_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, stream, cloud, n, sorted, cell_start, g, cov6, list + 2, list, far_ok,
                         far_list, list + 1, d_stats);
    if (far_ok)  // (the first cap: the radius the last cube tried has already failed)
      hipLaunchKernelGGL(gicp_cov_far_kernel, dim3(std::min(n, 256)), dim3(1024), 0, stream, cloud, n, cov6, (float)kCovFarLevel * g.h * kGridSafety, far_list,
                         list + 1);
    if (d_stats) {
      unsigned long long h[16];
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost);
      (void)hipFree(d_stats);
      fprintf(stderr, "[icpgpu] covariances of %d points (cells of %.3f m): %llu selected (%.1f levels, %.0f keys inside the certified radius, %.2f threshold probes, %.1f keys ranked "
                      "each), %llu dense corners collected again; handed on: %llu caps not settled, %llu whole grid searched\n", n, (double)g.h, h[0],
              (double)h[2] / (double)(h[0] ? h[0] : 1), (double)h[1] / (double)(h[0] ? h[0] : 1), (double)h[6] / (double)(h[0] ? h[0] : 1),
              (double)h[7] / (double)(h[0] ? h[0] : 1), h[3], h[4], h[5]);
      int h_lists[2] = {0, 0};
      (void)hipMemcpy(h_lists, list, sizeof(h_lists), hipMemcpyDeviceToHost);
      fprintf(stderr, "[icpgpu]   handed over: %d to the far-field kernel, %d to the streaming kernel (before the far-field kernel ran)\n", h_lists[1], h_lists[0]);
      fprintf(stderr, "[icpgpu]   points by the cube radius (in cells) that certified them: 1: %llu, 2: %llu, 4: %llu, 8: %llu, 16: %llu, 32: %llu, 64: %llu, more: %llu\n", h[8],
              h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
    }
    // (with the far-field kernel nothing is left over: it settles every point it is handed)
    if (!far_ok) hipLaunchKernelGGL(gicp_cov_kernel, dim3(std::min((n + 3) / 4, 128)), dim3(256), 0, stream, cloud, n, sorted, cell_start, g, cov6, list + 2, list);
  } else {
    hipLaunchKernelGGL(gicp_cov_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, cloud, n, sorted, cell_start, g, cov6, nullptr, nullptr);
  }
  hipLaunchKernelGGL(gicp_cov_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, cov6, (list && select) ? list : nullptr);
  return hipGetLastError();
}

// The covariances of a cloud WITHOUT a grid (the k-NN grid refuses a cloud whose densest cell of the largest table holds thousands of
// points): every finite point through the far-field kernel's whole-cloud passes.  n <= kGicpCovFarMost; list as above.
hipError_t launch_gicp_covariances_brute(const float4* cloud, int n, double* cov6, hipStream_t stream, int* list, bool list_counters_zero) {
  if (n <= 0) return hipSuccess;
  if (n > kGicpCovFarMost || !list) return hipErrorInvalidValue;
  if (!list_counters_zero) {
    hipError_t e = hipMemsetAsync(list, 0, 2 * sizeof(int), stream);
    if (e != hipSuccess) return e;
  }
  int* far_list = list + 2 + n;
  hipLaunchKernelGGL(gicp_cov_all_far_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cloud, n, cov6, far_list, list + 1);
  hipLaunchKernelGGL(gicp_cov_far_kernel, dim3(std::min(n, 1024)), dim3(1024), 0, stream, cloud, n, cov6, 0.05f, far_list, list + 1);
  hipLaunchKernelGGL(gicp_cov_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, cov6, list);
  return hipGetLastError();
}

hipError_t launch_gicp_mahalanobis(int n_s, const unsigned long long* keys, float thr, const Rot3d& R, const double* cov_s,
                                   const double* cov_t, double* maha6, hipStream_t stream) {
  if (n_s <= 0) return hipSuccess;
  hipLaunchKernelGGL(gicp_maha_kernel, dim3((n_s + 255) / 256), dim3(256), 0, stream, n_s, keys, thr, R, cov_s, cov_t, maha6);
  return hipGetLastError();
}

int gicp_quadratic_blocks(int n_s) {
  const int b = (n_s + 511) / 512;  // two correspondences per lane
  return b < 1 ? 1 : b > kGicpQuadBlocks ? kGicpQuadBlocks : b;
}
hipError_t launch_gicp_quadratic(const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                                 const Rot3d& R, const double* cov_s, const double* cov_t, double* partials, unsigned int* done,
                                 unsigned long long* host_out, unsigned long long seq, hipStream_t stream) {
  static const int stamps = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_TIMING"); return e && atoi(e) != 0 ? 1 : 0; }();
  hipLaunchKernelGGL(gicp_quadratic_kernel, dim3(gicp_quadratic_blocks(n_s)), dim3(256), 0, stream, src, n_s, tgt, keys, thr, R, cov_s,
                     cov_t, reinterpret_cast<DD*>(partials), done, host_out, seq, stamps);
  return hipGetLastError();
}

int gicp_direct_blocks(int n_s, int most) {
  // ICPGPU_GICP_BLOCKS (tuning): the most workgroups an evaluation may use.  64 until round 2; with one workgroup per 1024
  // points up to 256 a 200k x 200k registration takes 3.9 instead of 4.45 ms (12 instead of 15 us per evaluation: the
  // workgroups' share shrinks faster than the host's merge of their partial sums grows).  Keeping each lane's
  // correspondences in registers for the whole run of the server (gicp_server_kernel<true>) takes another 9-11 % off at
  // every size (launch_gicp_server picks it whenever a lane's share is one quad).
  static const int cap = [] {
    const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_BLOCKS");
    const int v = e ? std::atoi(e) : kGicpDirectBlocks;
    return v < 1 ? 1 : (v > kGicpDirectBlocks ? kGicpDirectBlocks : v);
  }();
  static const int per_block = [] {  // ICPGPU_GICP_PER_BLOCK (tuning): correspondences per workgroup of 256 lanes
    const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_PER_BLOCK");
    // 512 since round 6 (1024 before): two correspondences per lane halve a workgroup's accumulation (1.3 us of an evaluation's ~6),
    // and twice the answer lines cost the host little now that it validates and merges them as vectors -- the alignment stage of a
    // pipeline scan 1.24 -> 1.15 ms (256: 1.25, 384: 1.17, 640: 1.19, 768: 1.25; scripts/r6/perblock.sh).  The sums do not depend on
    // the split: they are exact sums rounded once.
    const int v = e ? std::atoi(e) : 512;
    return v < 256 ? 256 : (v > 4096 ? 4096 : v);
  }();
  int blocks = (n_s + per_block - 1) / per_block;
  if (blocks > cap) blocks = cap;
  if (blocks > most) blocks = most;  // (the caller's share of the chip: several contexts' servers must all be resident)
  if (blocks < 1) blocks = 1;
  return blocks;
}

hipError_t launch_gicp_cost_direct(int blocks, const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                                   const Xform& T, const Xform& base, const double* maha6, double* host_partials,
                                   unsigned long long* host_flags, unsigned long long seq, hipStream_t stream) {
  hipLaunchKernelGGL(gicp_cost_kernel, dim3(blocks), dim3(256), 0, stream, src, n_s, tgt, keys, thr, T, base,
                     maha6, host_partials, host_flags, seq);
  return hipGetLastError();
}

hipError_t launch_gicp_server(int blocks, const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                              const Xform& base, const double* maha6, double* host_partials, unsigned long long* host_flags,
                              unsigned int* cmd, unsigned int first_seq, unsigned int seq_hi, hipStream_t stream) {
  static const int stamps = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_TIMING"); return e && std::atoi(e) != 0 ? 1 : 0; }();
  // every lane's share fits one quad: up to 256 workgroups x 1024 points (ICPGPU_GICP_RESIDENT_MAX: tuning switch; 0 = never)
  static const int resident_max = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_RESIDENT_MAX"); return e ? std::atoi(e) : 262144; }();
  if ((long long)blocks * 1024 >= n_s && n_s <= resident_max)
    hipLaunchKernelGGL(gicp_server_kernel<true>, dim3(blocks), dim3(256), 0, stream, src, n_s, tgt, keys, thr, base, maha6,
                       host_partials, host_flags, cmd, first_seq, seq_hi, stamps);
  else
    hipLaunchKernelGGL(gicp_server_kernel<false>, dim3(blocks), dim3(256), 0, stream, src, n_s, tgt, keys, thr, base, maha6,
                       host_partials, host_flags, cmd, first_seq, seq_hi, stamps);
  return hipGetLastError();
}


// The device solver: one resident kernel per outer iteration (see gicp_solve_kernel).  `blocks` workgroups of 256 lanes, all
// co-resident; RESIDENT when every lane's share of the correspondences is one quad.
size_t gicp_solve_slot_bytes(int blocks) { return (size_t)2 * (size_t)blocks * kSolveGranules * 2 * sizeof(unsigned long long); }
int gicp_solve_out_granules() { return kSolveOut; }
hipError_t launch_gicp_solve(int blocks, const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                             const Xform& base, const float guess[16], const double* maha6, const double x0[6],
                             unsigned long long* slots, unsigned long long* host_out, unsigned long long seq0, int max_inner,
                             double gradient_tol, hipStream_t stream, unsigned long long* local_slots, unsigned long long* owner,
                             int xcc_want) {
  GicpSolveArgs A;
  A.src = src;
  A.n_s = n_s;
  A.tgt = tgt;
  A.keys = keys;
  A.thr = thr;
  A.base = base;
  for (int i = 0; i < 16; ++i) A.guess[i] = guess[i];
  A.maha6 = maha6;
  for (int i = 0; i < 6; ++i) A.x0[i] = x0[i];
  A.slots = slots;
  A.host_out = host_out;
  A.seq0 = seq0 & ~kMailboxReleaseBit;
  A.host_seq = seq0;
  A.max_inner = max_inner;
  A.gradient_tol = gradient_tol;
  A.workers = blocks;
  A.xcc_want = xcc_want;
  A.owner = owner;
  const bool resident = (long long)blocks * 1024 >= n_s;
  if (local_slots && owner && resident && blocks <= kSolveLocalBlocks) {  // one XCD: up to 32 workgroups (one per CU of the XCD)
    A.slots = local_slots;
    hipLaunchKernelGGL((gicp_solve_kernel<true, true>), dim3(8 * blocks), dim3(256), 0, stream, A);
  } else if (resident) {
    hipLaunchKernelGGL((gicp_solve_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, A);
  } else {
    hipLaunchKernelGGL((gicp_solve_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, A);
  }
  return hipGetLastError();
}
hipError_t launch_gicp_solve_batch(const GicpSolveItem* items, int n, int max_inner, double gradient_tol, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (n > kGicpSolveBatchMax) return hipErrorInvalidValue;
  GicpSolveBatchArgs B;
  int max_blocks = 0;
  for (int k = 0; k < n; ++k) {
    const GicpSolveItem& it = items[k];
    if ((long long)it.blocks * 1024 < it.n_s) return hipErrorInvalidValue;  // resident shares only (the caller launches the others singly)
    GicpSolveArgs& A = B.a[k];
    A.src = it.src;
    A.n_s = it.n_s;
    A.tgt = it.tgt;
    A.keys = it.keys;
    A.thr = it.thr;
    A.base = it.base;
    for (int i = 0; i < 16; ++i) A.guess[i] = it.guess[i];
    A.maha6 = it.maha6;
    for (int i = 0; i < 6; ++i) A.x0[i] = it.x0[i];
    A.slots = it.slots;
    A.host_out = it.host_out;
    A.seq0 = it.seq0 & ~kMailboxReleaseBit;
    A.host_seq = it.seq0;
    A.max_inner = max_inner;
    A.gradient_tol = gradient_tol;
    A.workers = it.blocks;
    A.xcc_want = 0;
    A.owner = nullptr;
    max_blocks = it.blocks > max_blocks ? it.blocks : max_blocks;
  }
  for (int k = n; k < kGicpSolveBatchMax; ++k) B.a[k] = B.a[0];
  hipLaunchKernelGGL(gicp_solve_batch_kernel, dim3(max_blocks, n), dim3(256), 0, stream, B);
  return hipGetLastError();
}
int gicp_solve_local_blocks() { return kSolveLocalBlocks; }
// host side of a granule: true and the value if it carries number `seq` and its own checksum
bool gicp_granule_read(const volatile unsigned long long* g, unsigned long long seq, double* value) {
  unsigned long long bits;
  if (!mailbox_read(g, seq, &bits)) return false;
  std::memcpy(value, &bits, sizeof bits);
  return true;
}

}  // namespace icpgpu
