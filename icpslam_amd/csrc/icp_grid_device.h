// icp_grid_device.h -- device helpers of the uniform-grid search used by icp_grid.hip (internal).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#include "icp_device.h"
#include "icp_kernels.h"

#ifndef ICPGPU_BALL_PRUNING
#define ICPGPU_BALL_PRUNING 1  // 0: walk whole cubes (the first version; for A/B measurements)
#endif

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
// __shfl_xor compiles to, and the compiler folds them into the consuming v_min (one instruction per step).
// Controls (GFX9 encoding): quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140.
template <int CTRL>
__device__ __forceinline__ unsigned int dpp_move(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}

// minimum of v over the 64 lanes, returned wave-uniform (an SGPR)
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
  v = min(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]: lane ^ 1
  v = min(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]: lane ^ 2
  v = min(v, dpp_move<0x141>(v));  // row_half_mirror: the other quad of each 8
  v = min(v, dpp_move<0x140>(v));  // row_mirror: the other half of each 16 -> every lane holds its row's minimum
  const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned int)__builtin_amdgcn_readlane((int)v, 16),
                     r2 = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(r0, r1), min(r2, r3));  // scalar unit
}

// minimum of v over each row of 16 lanes, left in every lane of the row
__device__ __forceinline__ unsigned int row16_min_u32(unsigned int v) {
  v = min(v, dpp_move<0xB1>(v));
  v = min(v, dpp_move<0x4E>(v));
  v = min(v, dpp_move<0x141>(v));
  return min(v, dpp_move<0x140>(v));
}

// value of v in another lane, per-lane source (ds_bpermute: the LDS crossbar, no memory access)
__device__ __forceinline__ int lane_get_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float lane_get_f(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

__device__ __forceinline__ float readlane_f(float v, int lane) {  // lane must be wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

struct LaneBest {
  unsigned long long key;  // (d2 bits << 32) | original target index
  float qx, qy, qz;        // the candidate itself (for the fused reduction)
};

// The same with the candidate's POSITION (byte offset into `sorted`) instead of its coordinates: one register instead of
// three through the cube search, and what nn_quad_kernel remembers of a point's neighbour for the next sweep (4 bytes per
// source point instead of 16).  The coordinates are fetched once, from the position, when the search is over.
struct LanePos {
  unsigned long long key;
  unsigned int pos;
};

// Both points are finite here (non-finite targets are never binned, non-finite queries never search), so d2 is a finite
// float or +inf, never NaN: its bit pattern orders like its value and no NaN guard is needed.
// (pos = the candidate's byte offset in `sorted`; the coordinate-keeping flavour ignores it)
__device__ __forceinline__ void consider(const float4& q, unsigned int pos, float px, float py, float pz, LaneBest& b) {
  const float d = dist2(q.x, q.y, q.z, px, py, pz);
  const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q.w);
  if (key < b.key) {
    b.key = key;
    b.qx = q.x;
    b.qy = q.y;
    b.qz = q.z;
  }
}
__device__ __forceinline__ void consider(const float4& q, unsigned int pos, float px, float py, float pz, LanePos& b) {
  const float d = dist2(q.x, q.y, q.z, px, py, pz);
  const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | __float_as_uint(q.w);
  if (key < b.key) {
    b.key = key;
    b.pos = pos;
  }
}
__device__ __forceinline__ void take_owner(LaneBest& b, int owner) {
  b.qx = readlane_f(b.qx, owner);
  b.qy = readlane_f(b.qy, owner);
  b.qz = readlane_f(b.qz, owner);
}
__device__ __forceinline__ void take_owner(LanePos& b, int owner) { b.pos = (unsigned int)__builtin_amdgcn_readlane((int)b.pos, owner); }

// wave-wide winner: key and candidate become wave-uniform; false when no lane holds a candidate.  The distance bits are
// reduced first (6 DPP steps); only when several lanes tie on the distance is a second reduction over the original
// indices needed (lowest index wins, the contract of DESIGN.md section 3).
template <class Best>
__device__ __forceinline__ bool merge_lanes(Best& b) {
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
  take_owner(b, owner);
  return true;
}

// Walk the cell rows whose (lo, len) sit in the lanes named by `mask` (a row = fixed y,z and a contiguous x run = ONE
// range of `sorted`), two rows per step.  Row bounds are read with v_readlane (the lane index is wave-uniform), so a
// row's base address is scalar.  Lanes past the end of a row re-read its last entry (same cache line, no extra traffic)
// instead of being masked: evaluating a target point twice cannot change an exact minimum, so there is no per-lane
// bounds test, and both loads of a step are in flight together.
template <class Best>
__device__ __forceinline__ void sweep_rows(const float4* __restrict__ sorted, int lo, int len, unsigned long long mask,
                                           unsigned int lane, float px, float py, float pz, Best& b,
                                           unsigned int* n_cand = nullptr) {
  while (mask) {
    const int ra = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const int alo = __builtin_amdgcn_readlane(lo, ra), alen = __builtin_amdgcn_readlane(len, ra);
    int blo = alo, blen = 1;  // no second row: row a's first entry again
    if (mask) {
      const int rb = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      blo = __builtin_amdgcn_readlane(lo, rb);
      blen = __builtin_amdgcn_readlane(len, rb);
    }
    if (n_cand) *n_cand += (unsigned int)(alen + (blo != alo || blen != 1 ? blen : 0));  // (counting runs only; wave-uniform)
    const unsigned int ia = (unsigned int)alo + min(lane, (unsigned int)(alen - 1));
    const unsigned int ib = (unsigned int)blo + min(lane, (unsigned int)(blen - 1));
    const float4 qa = sorted[ia];
    const float4 qb = sorted[ib];
    consider(qa, ia << 4, px, py, pz, b);
    consider(qb, ib << 4, px, py, pz, b);
    for (int k = 64; k < alen; k += 64) {  // long rows (dense cells near the sensor)
      const unsigned int i2 = (unsigned int)(alo + k) + min(lane, (unsigned int)(alen - 1 - k));
      consider(sorted[i2], i2 << 4, px, py, pz, b);
    }
    for (int k = 64; k < blen; k += 64) {
      const unsigned int i2 = (unsigned int)(blo + k) + min(lane, (unsigned int)(blen - 1 - k));
      consider(sorted[i2], i2 << 4, px, py, pz, b);
    }
  }
}

// As sweep_rows, but rows of at most 16 entries go FOUR at a time: 16 lanes per row, one load and one evaluation for the
// four (the cubes of the fall-back path over a sparse target are made of such rows: 9 rows of a cube cost 3 steps
// instead of 5).  Longer rows take the two-at-a-time path.  Measured: -10 % at 50k x 50k, -19 % at 5k x 5k, but +4 % at
// 200k x 200k, so it is a template switch (the mere presence of this code in the
// kernel costs the dense case 3 %) that the host picks by the target's cell population.
template <class Best>
__device__ __forceinline__ void sweep_rows_packed(const float4* __restrict__ sorted, int lo, int len, unsigned long long mask,
                                                  unsigned int lane, float px, float py, float pz, Best& b,
                                                  unsigned int* n_cand = nullptr) {
  unsigned long long shortm = mask & __ballot(len <= 16);
  const unsigned long long longm = mask & ~shortm;
  const unsigned int seg = lane >> 4, sub = lane & 15u;
  while (shortm) {
    const int r0 = __ffsll((long long)shortm) - 1;
    shortm &= shortm - 1;
    int r1 = r0, r2 = r0, r3 = r0;  // fewer than four rows left: row 0 again (harmless)
    if (shortm) {
      r1 = __ffsll((long long)shortm) - 1;
      shortm &= shortm - 1;
    }
    if (shortm) {
      r2 = __ffsll((long long)shortm) - 1;
      shortm &= shortm - 1;
    }
    if (shortm) {
      r3 = __ffsll((long long)shortm) - 1;
      shortm &= shortm - 1;
    }
    const int lo0 = __builtin_amdgcn_readlane(lo, r0), lo1 = __builtin_amdgcn_readlane(lo, r1),
              lo2 = __builtin_amdgcn_readlane(lo, r2), lo3 = __builtin_amdgcn_readlane(lo, r3);
    const int n0 = __builtin_amdgcn_readlane(len, r0), n1 = __builtin_amdgcn_readlane(len, r1),
              n2 = __builtin_amdgcn_readlane(len, r2), n3 = __builtin_amdgcn_readlane(len, r3);
    if (n_cand) *n_cand += (unsigned int)(n0 + (r1 != r0 ? n1 : 0) + (r2 != r0 ? n2 : 0) + (r3 != r0 ? n3 : 0));
    const int mylo = seg == 0 ? lo0 : seg == 1 ? lo1 : seg == 2 ? lo2 : lo3;
    const int mylen = seg == 0 ? n0 : seg == 1 ? n1 : seg == 2 ? n2 : n3;
    const unsigned int im = (unsigned int)mylo + min(sub, (unsigned int)(mylen - 1));
    consider(sorted[im], im << 4, px, py, pz, b);
  }
  sweep_rows(sorted, lo, len, longm, lane, px, py, pz, b, n_cand);
}

// The 2x2x2 octant of cells a point leans towards: lane `sel` in 0..3 gets one of its four cell rows.  Along each axis
// the point sits at fraction f of its cell and the octant reaches max(f, 1 - f) >= 1/2 cells beyond it on the nearer
// side, so every cell outside the octant is at least `margin` = the smallest of the three (in cells, 0.625 on average)
// away: a best distance <= 63/64 * margin * h found inside the octant is final.
__device__ __forceinline__ void octant_row(const int* __restrict__ cell_start, const GridDesc& g, float px, float py,
                                           float pz, int cx, int cy, int cz, int sel, int& lo, int& len, float& margin) {
  const float fx = (px - g.ox) * g.inv_h - (float)cx, fy = (py - g.oy) * g.inv_h - (float)cy, fz = (pz - g.oz) * g.inv_h - (float)cz;
  const int ax = cx + (fx < 0.5f ? -1 : 0), ay = cy + (fy < 0.5f ? -1 : 0), az = cz + (fz < 0.5f ? -1 : 0);
  margin = fminf(fminf(fmaxf(fx, 1.0f - fx), fmaxf(fy, 1.0f - fy)), fmaxf(fz, 1.0f - fz));
  const int x0 = max(ax, 0), x1 = min(ax + 1, g.nx - 1);
  const int yy = ay + (sel & 1), zz = az + ((sel >> 1) & 1);
  lo = 0;
  len = 0;
  if (x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
    const int row = zz * g.sz + yy * g.sy;
    lo = cell_start[row + x0];
    len = cell_start[row + x1 + 1] - lo;
  }
}

// cells^2 radius of the ball that must be searched when a target point is known at squared distance d2 (NaN or +inf:
// nothing known, no pruning).  See grow_cubes for the two safety terms.
__device__ __forceinline__ float grid_slack(const GridDesc& g) {
  return 0.03125f + (float)max(g.nx, max(g.ny, g.nz)) * (4.0f / 8388608.0f);
}
// v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the IEEE-exact expansions (14 and 10 instructions): these values only size
// search regions, every one of them inflated by 3 % + slack -- an ulp cannot decide anything.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float ball_cells_sq(const GridDesc& g, float d2) {
  const float ball = fast_sqrt(d2) * g.inv_h * 1.03125f + grid_slack(g);
  return d2 < __builtin_inff() ? ball * ball : __builtin_inff();
}

// octant_row for a point whose neighbour is known to lie within the ball of ball_sq (cells^2): rows of the octant
// outside the ball come back empty, the others trimmed to the ball's chord.
__device__ __forceinline__ void octant_row_in_ball(const int* __restrict__ cell_start, const GridDesc& g, float px, float py,
                                                   float pz, int cx, int cy, int cz, int sel, float ball_sq, int& lo,
                                                   int& len, float& margin) {
  const float ux = (px - g.ox) * g.inv_h, uy = (py - g.oy) * g.inv_h, uz = (pz - g.oz) * g.inv_h;
  const float fx = ux - (float)cx, fy = uy - (float)cy, fz = uz - (float)cz;
  const int ax = cx + (fx < 0.5f ? -1 : 0), ay = cy + (fy < 0.5f ? -1 : 0), az = cz + (fz < 0.5f ? -1 : 0);
  margin = fminf(fminf(fmaxf(fx, 1.0f - fx), fmaxf(fy, 1.0f - fy)), fmaxf(fz, 1.0f - fz));
  int x0 = max(ax, 0), x1 = min(ax + 1, g.nx - 1);
  const int yy = ay + (sel & 1), zz = az + ((sel >> 1) & 1);
  lo = 0;
  len = 0;
  if (x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
    const float dy = fmaxf(fmaxf((float)yy - uy, uy - (float)(yy + 1)), 0.f);
    const float dz = fmaxf(fmaxf((float)zz - uz, uz - (float)(zz + 1)), 0.f);
    const float rem = ball_sq - dy * dy - dz * dz;
    if (rem >= 0.f) {
      const float w = fast_sqrt(rem);
      x0 = (int)fmaxf(floorf(ux - w), (float)x0);
      x1 = (int)fminf(floorf(ux + w), (float)x1);
      if (x0 <= x1) {
        const int row = zz * g.sz + yy * g.sy;
        lo = cell_start[row + x0];
        len = cell_start[row + x1 + 1] - lo;
      }
    }
  }
}

// Cubes of Chebyshev radius 1, 2, 4, ... (capped at r_max) around cell (cx, cy, cz) until the best distance is provably
// inside the cube; px..cz are wave-uniform, and so is b on entry (the octant's winner, or kEmptyKey) and on return.
//
// Ball pruning: once some point is known at distance D (the octant's winner, then each cube's), the neighbour lies in
// the ball of radius D, and a cube of 5^3 or 9^3 cells is mostly outside it.  Each lane drops its cell row when the row's
// (y, z) slab is farther than D, and otherwise trims the row's x range to the chord of the ball: the walk then costs what
// the ball holds, not what the cube holds.  Everything is done in cell units on the SAME float coordinates the binning
// used; `slack` bounds what those can be off by (two roundings of a value below max(nx, ny, nz): < n * 2^-23 per point,
// query and target, three axes), and D itself is inflated by 1/32 against the rounding of dist2.  The pruning only
// removes cells that cannot hold a point closer than D; the certification test is unchanged.
template <bool PACK_SHORT_ROWS, class Best>
__device__ __forceinline__ bool grow_cubes(const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                           const GridDesc& g, float px, float py, float pz, int cx, int cy, int cz,
                                           unsigned int lane, Best& b, int rho_first = 1, unsigned int* n_cand = nullptr) {
  const float fx = (px - g.ox) * g.inv_h, fy = (py - g.oy) * g.inv_h, fz = (pz - g.oz) * g.inv_h;
  const float slack = grid_slack(g);
  for (int rho = rho_first;; rho = min(2 * rho, g.r_max)) {
    const int side = 2 * rho + 1, nrows = side * side;
    const int x0 = max(cx - rho, 0), x1 = min(cx + rho, g.nx - 1);
    const float inv_side = __builtin_amdgcn_rcpf((float)side);  // (r + 1/2) / side is never within 1/(2 side) of an integer
    // radius of the ball in cells (+inf while nothing has been found: no pruning)
    const float best = __uint_as_float((unsigned int)(b.key >> 32));  // kEmptyKey reads as a NaN
    const float ball = fast_sqrt(best) * g.inv_h * 1.03125f + slack;
    const float ball_sq = (ICPGPU_BALL_PRUNING && best < __builtin_inff()) ? ball * ball : __builtin_inff();
    for (int rb = 0; rb < nrows; rb += 64) {
      const int r = rb + (int)lane;  // lane -> one cell row of the cube; (y, z) by an exact float reciprocal
      const int zr = (int)(((float)r + 0.5f) * inv_side), yr = r - zr * side;
      const int yy = cy + yr - rho, zz = cz + zr - rho;
      int lo = 0, len = 0;
      if (r < nrows && x0 <= x1 && yy >= 0 && yy < g.ny && zz >= 0 && zz < g.nz) {
        // distance (in cells) from the query to the slab [yy, yy+1) x [zz, zz+1); 0 inside it
        const float dy = fmaxf(fmaxf((float)yy - fy, fy - (float)(yy + 1)), 0.f);
        const float dz = fmaxf(fmaxf((float)zz - fz, fz - (float)(zz + 1)), 0.f);
        const float rem = ball_sq - dy * dy - dz * dz;
        if (rem >= 0.f) {
          const float w = fast_sqrt(rem);
          const int xa = (int)fmaxf(floorf(fx - w), (float)x0), xb = (int)fminf(floorf(fx + w), (float)x1);
          if (xa <= xb) {
            const int row = zz * g.sz + yy * g.sy;
            lo = cell_start[row + xa];
            len = cell_start[row + xb + 1] - lo;
          }
        }
      }
      if constexpr (PACK_SHORT_ROWS) sweep_rows_packed(sorted, lo, len, __ballot(len > 0), lane, px, py, pz, b, n_cand);
      else sweep_rows(sorted, lo, len, __ballot(len > 0), lane, px, py, pz, b, n_cand);
    }
    const bool any = merge_lanes(b);
    const float safe = (float)rho * g.h * kGridSafety;
    if (any && __uint_as_float((unsigned int)(b.key >> 32)) <= safe * safe) return true;
    if (rho >= g.r_max) return false;
  }
}

}  // namespace icpgpu
