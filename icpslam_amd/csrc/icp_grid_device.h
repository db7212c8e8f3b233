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

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
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

// wave-wide winner: key and candidate broadcast to every lane; returns false when no lane holds a candidate
__device__ __forceinline__ bool merge_lanes(LaneBest& b) {
  const unsigned long long wbest = wave_min_u64(b.key);
  const unsigned long long owner = __ballot(b.key == wbest && wbest != kEmptyKey);
  if (!owner) return false;
  const int ol = __ffsll((long long)owner) - 1;
  b.key = wbest;
  b.qx = __shfl(b.qx, ol, 64);
  b.qy = __shfl(b.qy, ol, 64);
  b.qz = __shfl(b.qz, ol, 64);
  return true;
}

// walk the non-empty rows among the 64 (lo, len) pairs held by the lanes, two rows per step
__device__ __forceinline__ void sweep_rows(const float4* __restrict__ sorted, int lo, int len, int lane, float px, float py,
                                           float pz, LaneBest& b) {
  unsigned long long mask = __ballot(len > 0);
  while (mask) {  // wave-uniform
    const int ra = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const int alo = __shfl(lo, ra, 64), alen = __shfl(len, ra, 64);
    int blo = 0, blen = 0;
    if (mask) {
      const int rb = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      blo = __shfl(lo, rb, 64);
      blen = __shfl(len, rb, 64);
    }
    float4 qa, qb;
    const bool va = lane < alen, vb = lane < blen;
    if (va) qa = sorted[alo + lane];
    if (vb) qb = sorted[blo + lane];
    if (va) consider(qa, px, py, pz, b);
    if (vb) consider(qb, px, py, pz, b);
    for (int k = 64 + lane; k < alen; k += 64) consider(sorted[alo + k], px, py, pz, b);  // long rows
    for (int k = 64 + lane; k < blen; k += 64) consider(sorted[blo + k], px, py, pz, b);
  }
}

// Whole-wave search for ONE query: cubes of Chebyshev radius rho_start, 2*rho_start, ... (capped at r_max) around cell
// (cx, cy, cz) until the best distance is provably inside the cube.  On return b holds the wave-uniform winner.
__device__ __forceinline__ bool grow_search(const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                            const GridDesc& g, float px, float py, float pz, int cx, int cy, int cz,
                                            int rho_start, int lane, LaneBest& b) {
  for (int rho = min(rho_start, g.r_max);; rho = min(2 * rho, g.r_max)) {
    const int side = 2 * rho + 1, nrows = side * side;
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.nx - 1);
    const float inv_side = 1.0f / (float)side;
    for (int rb = 0; rb < nrows; rb += 64) {
      const int r = rb + lane;  // lane -> one cell row of the cube; (y, z) by an exact float reciprocal
      const int zr = (int)(((float)r + 0.5f) * inv_side), yr = r - zr * side;
      const int yy = cy + yr - rho, zz = cz + zr - rho;
      int lo = 0, len = 0;
      if (r < nrows && x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
        const int row = (zz * g.ny + yy) * g.nx;
        lo = cell_start[row + x0];
        len = cell_start[row + x1 + 1] - lo;
      }
      sweep_rows(sorted, lo, len, lane, px, py, pz, b);
    }
    const bool any = merge_lanes(b);
    const float safe = (float)rho * g.h * kGridSafety;
    if (any && __uint_as_float((unsigned int)(b.key >> 32)) <= safe * safe) return true;
    if (rho >= g.r_max) return false;
  }
}

}  // namespace icpgpu
