// icp_grid_device.h -- device helpers of the uniform-grid search used by icp_grid.hip (internal).
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

// ---- wave-level primitives ------------------------------------------------------------------------------------------
// DPP (data-parallel primitive) lane moves stay inside the VALU: no LDS round trip like ds_bpermute, which is what
// __shfl_xor compiles to.  Controls (GFX9 encoding): quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140,
// row_bcast15 0x142 (lane 15 of every row -> the next row), row_bcast31 0x143 (lane 31 -> rows 2, 3).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_move(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// minimum of v over the 64 lanes, returned wave-uniform (an SGPR)
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
  v = min(v, dpp_move<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]: lane ^ 1
  v = min(v, dpp_move<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]: lane ^ 2
  v = min(v, dpp_move<0x141, 0xF>(v));  // row_half_mirror: the other quad of each 8
  v = min(v, dpp_move<0x140, 0xF>(v));  // row_mirror: the other half of each 16 -> every lane holds its row's minimum
  v = min(v, dpp_move<0x142, 0xA>(v));  // row_bcast15 into rows 1 and 3
  v = min(v, dpp_move<0x143, 0xC>(v));  // row_bcast31 into rows 2 and 3 -> lane 63 holds the wave minimum
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ float readlane_f(float v, int lane) {  // lane must be wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

struct LaneBest {
  unsigned long long key;  // (d2 bits << 32) | original target index
  float qx, qy, qz;        // the candidate itself (for the fused reduction)
};

// Both points are finite here (non-finite targets are never binned, non-finite queries never search), so d2 is a finite
// float or +inf, never NaN: its bit pattern orders like its value and no NaN guard is needed.
__device__ __forceinline__ void consider(const float4& q, float px, float py, float pz, bool valid, LaneBest& b) {
  const float d = dist2(q.x, q.y, q.z, px, py, pz);
  const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q.w);
  if (valid && key < b.key) {
    b.key = key;
    b.qx = q.x;
    b.qy = q.y;
    b.qz = q.z;
  }
}

// wave-wide winner: key and candidate become wave-uniform; false when no lane holds a candidate.  The distance bits are
// reduced first (6 DPP steps); only when several lanes tie on the distance is a second reduction over the original
// indices needed (lowest index wins, the contract of DESIGN.md section 3).
__device__ __forceinline__ bool merge_lanes(LaneBest& b) {
  const unsigned int dbits = (unsigned int)(b.key >> 32), idx = (unsigned int)b.key;
  const unsigned int dmin = wave_min_u32(dbits);
  if (dmin == 0xFFFFFFFFu) return false;  // every lane still holds kEmptyKey
  unsigned long long tied = __ballot(dbits == dmin);
  unsigned int imin;
  if ((tied & (tied - 1)) == 0) {
    imin = (unsigned int)__builtin_amdgcn_readlane((int)idx, __ffsll((long long)tied) - 1);
  } else {
    imin = wave_min_u32(dbits == dmin ? idx : 0xFFFFFFFFu);
    tied = __ballot(dbits == dmin && idx == imin);
  }
  const int owner = __ffsll((long long)tied) - 1;
  b.key = ((unsigned long long)dmin << 32) | imin;
  b.qx = readlane_f(b.qx, owner);
  b.qy = readlane_f(b.qy, owner);
  b.qz = readlane_f(b.qz, owner);
  return true;
}

// Walk the cell rows whose (lo, len) sit in the lanes named by `mask` (a row = fixed y,z and a contiguous x run = ONE
// range of `sorted`), two rows per step.  Row bounds are read with v_readlane (the lane index is wave-uniform), both
// loads are issued unconditionally at clamped positions so that they are in flight together, and lanes past the end of
// a row are masked out in the comparison instead.
__device__ __forceinline__ void sweep_rows(const float4* __restrict__ sorted, int lo, int len, unsigned long long mask,
                                           int lane, float px, float py, float pz, LaneBest& b) {
  while (mask) {
    const int ra = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const int alo = __builtin_amdgcn_readlane(lo, ra), alen = __builtin_amdgcn_readlane(len, ra);
    int blo = alo, blen = 0;
    if (mask) {
      const int rb = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      blo = __builtin_amdgcn_readlane(lo, rb);
      blen = __builtin_amdgcn_readlane(len, rb);
    }
    const float4 qa = sorted[alo + min(lane, alen - 1)];
    const float4 qb = sorted[blo + min(lane, max(blen - 1, 0))];
    consider(qa, px, py, pz, lane < alen, b);
    consider(qb, px, py, pz, lane < blen, b);
    for (int k = 64; k < alen; k += 64) {  // long rows (dense cells close to the sensor)
      const float4 q = sorted[alo + min(k + lane, alen - 1)];
      consider(q, px, py, pz, k + lane < alen, b);
    }
    for (int k = 64; k < blen; k += 64) {
      const float4 q = sorted[blo + min(k + lane, blen - 1)];
      consider(q, px, py, pz, k + lane < blen, b);
    }
  }
}

// The 2x2x2 octant of cells a point leans towards: lane `sel` in 0..3 gets one of its four cell rows.  Every cell outside
// the octant is at least h/2 away from the point, so a best distance <= 63/64 * h/2 found inside it is final.
__device__ __forceinline__ void octant_row(const int* __restrict__ cell_start, const GridDesc& g, float px, float py,
                                           float pz, int cx, int cy, int cz, int sel, int& lo, int& len) {
  const float fx = (px - g.ox) * g.inv_h - (float)cx, fy = (py - g.oy) * g.inv_h - (float)cy, fz = (pz - g.oz) * g.inv_h - (float)cz;
  const int ax = cx + (fx < 0.5f ? -1 : 0), ay = cy + (fy < 0.5f ? -1 : 0), az = cz + (fz < 0.5f ? -1 : 0);
  const int x0 = max(ax, 0), x1 = min(ax + 1, g.nx - 1);
  const int yy = ay + (sel & 1), zz = az + ((sel >> 1) & 1);
  lo = 0;
  len = 0;
  if (x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
    const int row = (zz * g.ny + yy) * g.nx;
    lo = cell_start[row + x0];
    len = cell_start[row + x1 + 1] - lo;
  }
}

// Cubes of Chebyshev radius 1, 2, 4, ... (capped at r_max) around cell (cx, cy, cz) until the best distance is provably
// inside the cube; px..cz are wave-uniform.  On return b holds the wave-uniform winner.
__device__ __forceinline__ bool grow_cubes(const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                           const GridDesc& g, float px, float py, float pz, int cx, int cy, int cz,
                                           int lane, LaneBest& b) {
  for (int rho = 1;; rho = min(2 * rho, g.r_max)) {
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
      sweep_rows(sorted, lo, len, __ballot(len > 0), lane, px, py, pz, b);
    }
    const bool any = merge_lanes(b);
    const float safe = (float)rho * g.h * kGridSafety;
    if (any && __uint_as_float((unsigned int)(b.key >> 32)) <= safe * safe) return true;
    if (rho >= g.r_max) return false;
  }
}

}  // namespace icpgpu
