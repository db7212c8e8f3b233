// icp_grid_device.h -- device helpers of the uniform-grid search shared by icp_grid.hip and icp_group.hip (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#include "icp_device.h"
#include "icp_kernels.h"

namespace icpgpu {

__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// Cell coordinates: the SAME float expression bins targets and queries, and floor((x - o) * inv_h) is monotone in x,
// which is what the early-exit proof needs.
__device__ __forceinline__ void cell_of(const GridDesc& g, float x, float y, float z, int& cx, int& cy, int& cz) {
  const float fx = floorf((x - g.ox) * g.inv_h), fy = floorf((y - g.oy) * g.inv_h), fz = floorf((z - g.oz) * g.inv_h);
  const float lim = 1048576.0f;
  cx = (int)fminf(fmaxf(fx, -lim), lim);
  cy = (int)fminf(fmaxf(fy, -lim), lim);
  cz = (int)fminf(fmaxf(fz, -lim), lim);
}

// The search helpers are written for a GROUP of W lanes (W = 64: one query per wave; W = 32: two queries per wave, one
// per half).  gl = lane index inside the group, gshift = bit offset of the group inside the wave's 64-bit ballot.
template <int W>
__device__ __forceinline__ unsigned long long group_bits(unsigned long long ballot, int gshift) {
  if constexpr (W == 64) return ballot;
  else return (ballot >> gshift) & ((1ull << W) - 1ull);
}

template <int W>
__device__ __forceinline__ unsigned long long group_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, W);
    v = o < v ? o : v;
  }
  return v;
}

struct LaneBest {
  unsigned long long key;  // (d2 bits << 32) | original target index
  float qx, qy, qz;        // the candidate itself (for the fused reduction)
};

__device__ __forceinline__ void consider(const float4& q, float px, float py, float pz, LaneBest& b) {
  const float d = dist2(q.x, q.y, q.z, px, py, pz);
  const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q.w);
  if (d == d && key < b.key) {  // NaN never wins
    b.key = key;
    b.qx = q.x;
    b.qy = q.y;
    b.qz = q.z;
  }
}

// group-wide winner: key and candidate broadcast to every lane of the group; false when no lane holds a candidate
template <int W>
__device__ __forceinline__ bool merge_lanes(LaneBest& b, int gshift) {
  const unsigned long long gbest = group_min_u64<W>(b.key);
  const unsigned long long owner = group_bits<W>(__ballot(b.key == gbest && gbest != kEmptyKey), gshift);
  if (!owner) return false;
  const int ol = __ffsll((long long)owner) - 1;
  b.key = gbest;
  b.qx = __shfl(b.qx, ol, W);
  b.qy = __shfl(b.qy, ol, W);
  b.qz = __shfl(b.qz, ol, W);
  return true;
}

// walk the non-empty rows among the W (lo, len) pairs held by the group's lanes, two rows per step
template <int W>
__device__ __forceinline__ void sweep_rows(const float4* __restrict__ sorted, int lo, int len, int gl, int gshift, float px,
                                           float py, float pz, LaneBest& b) {
  unsigned long long mask = group_bits<W>(__ballot(len > 0), gshift);
  while (mask) {  // uniform within the group
    const int ra = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const int alo = __shfl(lo, ra, W), alen = __shfl(len, ra, W);
    int blo = 0, blen = 0;
    if (mask) {
      const int rb = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      blo = __shfl(lo, rb, W);
      blen = __shfl(len, rb, W);
    }
    float4 qa, qb;
    const bool va = gl < alen, vb = gl < blen;
    if (va) qa = sorted[alo + gl];
    if (vb) qb = sorted[blo + gl];
    if (va) consider(qa, px, py, pz, b);
    if (vb) consider(qb, px, py, pz, b);
    for (int k = W + gl; k < alen; k += W) consider(sorted[alo + k], px, py, pz, b);  // long rows
    for (int k = W + gl; k < blen; k += W) consider(sorted[blo + k], px, py, pz, b);
  }
}

// Group-wide search for ONE query.  Stage 0 (only when rho_start == 1): the 2x2x2 cells of the octant the query leans
// towards -- every excluded cell is at least h/2 away, so a best distance <= 63/64 * h/2 is final; this settles most
// points of a converging alignment with 4 rows instead of 9 and ~30 % of the candidates.  Then cubes of Chebyshev radius
// rho_start, 2*rho_start, ... (capped at r_max) around cell (cx, cy, cz) until the best distance is provably inside the
// cube.  On return b holds the group-uniform winner.
template <int W>
__device__ __forceinline__ bool grow_search(const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                            const GridDesc& g, float px, float py, float pz, int cx, int cy, int cz,
                                            int rho_start, int gl, int gshift, LaneBest& b) {
  if (rho_start == 1) {
    const float fx = (px - g.ox) * g.inv_h - (float)cx, fy = (py - g.oy) * g.inv_h - (float)cy, fz = (pz - g.oz) * g.inv_h - (float)cz;
    const int ax = cx + (fx < 0.5f ? -1 : 0), ay = cy + (fy < 0.5f ? -1 : 0), az = cz + (fz < 0.5f ? -1 : 0);
    const int x0 = max(ax, 0), x1 = min(ax + 1, g.nx - 1);
    const int yy = ay + (gl & 1), zz = az + ((gl >> 1) & 1);  // lanes 0..3 of the group -> the four rows
    int lo = 0, len = 0;
    if (gl < 4 && x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
      const int row = (zz * g.ny + yy) * g.nx;
      lo = cell_start[row + x0];
      len = cell_start[row + x1 + 1] - lo;
    }
    sweep_rows<W>(sorted, lo, len, gl, gshift, px, py, pz, b);
    const bool any = merge_lanes<W>(b, gshift);
    const float safe = 0.5f * g.h * kGridSafety;
    if (any && __uint_as_float((unsigned int)(b.key >> 32)) <= safe * safe) return true;
  }
  for (int rho = min(rho_start, g.r_max);; rho = min(2 * rho, g.r_max)) {
    const int side = 2 * rho + 1, nrows = side * side;
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.nx - 1);
    const float inv_side = 1.0f / (float)side;
    for (int rb = 0; rb < nrows; rb += W) {
      const int r = rb + gl;  // lane -> one cell row of the cube; (y, z) by an exact float reciprocal
      const int zr = (int)(((float)r + 0.5f) * inv_side), yr = r - zr * side;
      const int yy = cy + yr - rho, zz = cz + zr - rho;
      int lo = 0, len = 0;
      if (r < nrows && x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
        const int row = (zz * g.ny + yy) * g.nx;
        lo = cell_start[row + x0];
        len = cell_start[row + x1 + 1] - lo;
      }
      sweep_rows<W>(sorted, lo, len, gl, gshift, px, py, pz, b);
    }
    const bool any = merge_lanes<W>(b, gshift);
    const float safe = (float)rho * g.h * kGridSafety;
    if (any && __uint_as_float((unsigned int)(b.key >> 32)) <= safe * safe) return true;
    if (rho >= g.r_max) return false;
  }
}

}  // namespace icpgpu
