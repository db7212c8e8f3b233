// icp_tile.hip -- exact nearest neighbour through the uniform grid ON THE MATRIX CORES (rows a1 / a2 of SURVEY.md section 8(a):
// the correspondence search PCL's IterativeClosestPoint runs per iteration, reached from
// /root/reference/src/icpslam/icp_odometer.cpp:198 and src/icpslam/octree_mapper.cpp:114).
//
// icp_grid.hip searches the grid point by point (a 2 x 2 x 2 octant of cells per source, ball-pruned cubes after it): ~150
// vector instructions per source, 2.2 ps per candidate pair.  icp_brute_bf16.hip settles a pair in 0.03 ps -- one bf16 MFMA
// gives certified lower bounds of 32 x 32 distances, sixteen v_min3 fold them -- but offers every target to every source.  This
// kernel feeds the second machine with the first one's index:
//
//   a workgroup takes 256 sources that are neighbours in space (Morton order of their cells, launch_morton_order);
//   every source knows a radius its neighbour must lie within: the distance to the neighbour it found in the previous sweep
//   (whatever the transform is now, that is a target point), capped by the acceptance threshold (a neighbour beyond it is
//   rejected anyway: PCL's max correspondence distance);
//   the union of those balls gives a box of grid cells; the target points of those cells -- (rows of cells) x (contiguous
//   x range): a list of segments of the grid's sorted copy -- are streamed through LDS tiles, and every tile goes through the
//   bf16 lower-bound filter + exact re-check of icp_brute_bf16.hip.
//
// A converged sweep at 200k x 200k offers ~2 000 candidates to each source instead of 200 000 (or the octant's ~150 at 70 times
// the price per pair).  Exactness: the box holds every cell that holds a point within a source's radius (the binning
// expression is monotone and the box is the float-rounded ball, inflated); inside the box the search is the brute-force
// kernel's, whose bound is certified (header of icp_brute_bf16.hip; tau is small here: all candidates are within metres of the
// centre).  The key a source ends with is its exact nearest neighbour if that neighbour lies within the threshold, in contract
// arithmetic with the lowest original index among ties -- what nn_quad_kernel returns -- and empty otherwise.
#include <hip/hip_runtime.h>

#include "icp_env.h"

#include <cstdio>
#include <cstdlib>
#include <math.h>

#include "icp_device.h"
#include "icp_grid_device.h"
#include "icp_kernels.h"

namespace icpgpu {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float kTauBf16 = 1.220703125e-04f;  // 2^-13 (icp_brute_bf16.hip's error budget)
constexpr float kTinyScale = 1e-18f;
constexpr int TS_BLOCK = 256;   // 4 waves x 64 sources
constexpr int TS_G = 2;         // groups of 32 sources per wave
constexpr int TS_TILE = 512;    // candidates per LDS tile
constexpr int TS_ROWS = 1024;   // segments (cell rows) listed at a time

__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
  const floatx2 f = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ unsigned int dup_lo(unsigned int p) { return __builtin_amdgcn_perm(p, p, 0x01000100u); }
__device__ __forceinline__ unsigned int dup_hi(unsigned int p) { return __builtin_amdgcn_perm(p, p, 0x03020302u); }

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}

__global__ __launch_bounds__(TS_BLOCK) void nn_tile_kernel(const float4* __restrict__ src_morton, int n_q, Xform T,
                                                           const float4* __restrict__ sorted,
                                                           const int* __restrict__ cell_start, GridDesc g,
                                                           const float4* __restrict__ tgt, int n_t, float thr,
                                                           const unsigned long long* __restrict__ seed,
                                                           float4* __restrict__ prev, int use_prev,
                                                           unsigned long long* __restrict__ keys, int splits,
                                                           unsigned long long* __restrict__ stats) {
  constexpr int G = TS_G, TILE = TS_TILE, BLOCK = TS_BLOCK;
  __shared__ float4 tile[TILE];
  __shared__ uint4 aop[2 * TILE];
  __shared__ int seg_start[TS_ROWS];
  __shared__ int seg_pref[TS_ROWS + 1];
  __shared__ float s_box[BLOCK / 64][6];
  __shared__ float s_pmax[BLOCK / 64];
  __shared__ int s_wsum[BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int q0 = (blockIdx.x * (BLOCK / 64) + wave) * (32 * G);
  // (diagnostics, stats != nullptr: shader cycles per phase, summed over workgroups by wave 0)
  unsigned long long tk0 = stats ? __builtin_amdgcn_s_memtime() : 0ull, t_pre = 0, t_rows = 0, t_fill = 0, t_steps = 0;

  // ---- the workgroup's sources, their radii, the box of their balls ---------------------------------------------------------
  float px[G], py[G], pz[G], r2[G], bqx[G], bqy[G], bqz[G];
  int orig[G];
  bool valid[G];
  unsigned long long best[G];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const int k = q0 + gi * 32 + col;
    const float4 s = src_morton[min(k, n_q - 1)];
    xform_point(T, s.x, s.y, s.z, px[gi], py[gi], pz[gi]);
    orig[gi] = __float_as_int(s.w);
    valid[gi] = k < n_q && finite3(px[gi], py[gi], pz[gi]);
    best[gi] = kEmptyKey;
    bqx[gi] = bqy[gi] = bqz[gi] = __builtin_nanf("");
    r2[gi] = thr;  // a neighbour beyond the acceptance threshold is rejected anyway
    if ((use_prev & 1) && valid[gi]) {
      // (one split per group) the neighbour of the previous sweep as a POINT, stored at the source's Morton position: one
      // coalesced read instead of the key's two dependent gathers; .w = its original target index
      const float4 t = prev[min(k, n_q - 1)];
      const float e = dist2(t.x, t.y, t.z, px[gi], py[gi], pz[gi]);
      if (e <= thr) {
        best[gi] = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(t.w);
        r2[gi] = e;
        bqx[gi] = t.x; bqy[gi] = t.y; bqz[gi] = t.z;
      }
    } else if (seed && valid[gi]) {
      const unsigned int j = (unsigned int)seed[orig[gi]];
      if (j < (unsigned int)n_t) {
        const float4 t = tgt[j];
        const float e = dist2(t.x, t.y, t.z, px[gi], py[gi], pz[gi]);
        if (e <= thr) {  // the old neighbour is a target point: a candidate, and a ball the new neighbour lies in
          best[gi] = ((unsigned long long)__float_as_uint(e) << 32) | j;
          r2[gi] = e;
          bqx[gi] = t.x; bqy[gi] = t.y; bqz[gi] = t.z;
        }
      }
    }
    if (valid[gi] && r2[gi] >= 0.f) {
      // the ball's box in float: radius inflated for the rounding of the square root, the subtraction and the distance itself
      const float r = sqrtf(r2[gi]) * 1.001f + 1e-30f;
      const float ex = r + fabsf(px[gi]) * 1e-6f, ey = r + fabsf(py[gi]) * 1e-6f, ez = r + fabsf(pz[gi]) * 1e-6f;
      lo[0] = fminf(lo[0], px[gi] - ex); hi[0] = fmaxf(hi[0], px[gi] + ex);
      lo[1] = fminf(lo[1], py[gi] - ey); hi[1] = fmaxf(hi[1], py[gi] + ey);
      lo[2] = fminf(lo[2], pz[gi] - ez); hi[2] = fmaxf(hi[2], pz[gi] + ez);
    } else {
      valid[gi] = false;  // (a negative threshold accepts nothing)
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      s_box[wave][a] = lo[a];
      s_box[wave][3 + a] = hi[a];
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = s_box[0][a];
    hi[a] = s_box[0][3 + a];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) {
      lo[a] = fminf(lo[a], s_box[w][a]);
      hi[a] = fmaxf(hi[a], s_box[w][3 + a]);
    }
  }
  // cells of the box (cell_of is the binning's own expression, monotone per axis: every point between lo and hi lands between
  // their cells); clamped to the grid; an empty intersection leaves the seeds as they are
  int c0x, c0y, c0z, c1x, c1y, c1z;
  cell_of(g, lo[0], lo[1], lo[2], c0x, c0y, c0z);
  cell_of(g, hi[0], hi[1], hi[2], c1x, c1y, c1z);
  const bool any = lo[0] <= hi[0] && c1x >= 0 && c1y >= 0 && c1z >= 0 && c0x < g.nx && c0y < g.ny && c0z < g.nz;
  c0x = max(c0x, 0); c0y = max(c0y, 0); c0z = max(c0z, 0);
  c1x = min(c1x, g.nx - 1); c1y = min(c1y, g.ny - 1); c1z = min(c1z, g.nz - 1);
  const int rows_y = any ? c1y - c0y + 1 : 0, rows_z = any ? c1z - c0z + 1 : 0;
  const int n_rows = rows_y * rows_z;

  // ---- the filter's source side (icp_brute_bf16.hip): centre, P, the B operands ------------------------------------------------
  const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
  uint4 bop[G];
  float p2[G], bound[G], pmax2 = 0.f;
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const float ux = px[gi] - cx, uy = py[gi] - cy, uz = pz[gi] - cz;
    p2[gi] = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    if (valid[gi]) pmax2 = fmaxf(pmax2, p2[gi]);
    const float Ux = -2.0f * ux, Uy = -2.0f * uy, Uz = -2.0f * uz;
    const unsigned int hxy = pack_bf16(Ux, Uy), lxy = pack_bf16(Ux - bf16_lo(hxy), Uy - bf16_hi(hxy));
    const unsigned int hz = pack_bf16(Uz, 1.0f), lz = pack_bf16(Uz - bf16_lo(hz), 0.0f);
    const unsigned int dx = (hxy & 0xffffu) | (lxy << 16), dy = (hxy >> 16) | (lxy & 0xffff0000u);
    const unsigned int dz = (hz & 0xffffu) | (lz << 16);
    bop[gi] = half ? make_uint4(dz, dz, 0x3F803F80u, 0u) : make_uint4(dx, dx, dy, dy);
    // everything that can beat or tie what is known (or be accepted at all) has |q - p|^2 - |u|^2 <= r2 - |u|^2, widened for
    // the rounding of p2 and of this subtraction; a source that takes no part never asks for an exact evaluation
    bound[gi] = valid[gi] ? __builtin_fmaf(r2[gi] + p2[gi], 9.5367431640625e-07f, r2[gi] - p2[gi]) : -INFINITY;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pmax2 = fmaxf(pmax2, __shfl_xor(pmax2, off, 64));
  if (lane == 0) s_pmax[wave] = pmax2;
  __syncthreads();
  pmax2 = s_pmax[0];
#pragma unroll
  for (int w = 1; w < BLOCK / 64; ++w) pmax2 = fmaxf(pmax2, s_pmax[w]);

  if (stats) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_pre = t - tk0; tk0 = t; }
  float minus_inf;  // (opaque to the compiler: see icp_brute_bf16.hip)
  asm volatile("v_mov_b32 %0, 0xff800000" : "=v"(minus_inf));
  unsigned long long n_cand = 0;

  // ---- the box's rows, TS_ROWS at a time: segments of the sorted copy, their running total, this workgroup's share ---------------
  for (int row0 = 0; row0 < n_rows; row0 += TS_ROWS) {
    const int rows = min(TS_ROWS, n_rows - row0);
    __syncthreads();  // (the previous chunk's lists are no longer read)
    int mine[TS_ROWS / BLOCK], local = 0;
#pragma unroll
    for (int u = 0; u < TS_ROWS / BLOCK; ++u) {
      const int r = (int)threadIdx.x * (TS_ROWS / BLOCK) + u;  // consecutive rows per thread: a local running sum
      int len = 0;
      if (r < rows) {
        const int rr = row0 + r;
        const int yy = c0y + rr % rows_y, zz = c0z + rr / rows_y;
        const int base = zz * g.sz + yy * g.sy;
        const int a = cell_start[base + c0x];
        len = cell_start[base + c1x + 1] - a;
        seg_start[r] = a;
      }
      mine[u] = local;
      local += len;
    }
    const int incl = wave_incl_scan(local, lane);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int before = incl - local, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
      if (w < wave) before += s_wsum[w];
      total += s_wsum[w];
    }
#pragma unroll
    for (int u = 0; u < TS_ROWS / BLOCK; ++u) {
      const int r = (int)threadIdx.x * (TS_ROWS / BLOCK) + u;
      if (r < rows) seg_pref[r] = before + mine[u];
    }
    if (threadIdx.x == 0) seg_pref[rows] = total;
    // candidates [k0, k1) of this chunk are this workgroup's (blockIdx.y of `splits`), in steps of 32
    const int per = (((total + splits - 1) / splits) + 31) & ~31;
    const int k0 = min(total, (int)blockIdx.y * per), k1 = min(total, k0 + per);
    n_cand += (unsigned long long)(k1 - k0);
    if (stats) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_rows += t - tk0; tk0 = t; }

    // the candidates of a tile are gathered one tile ahead (the binary search over the row list and the read of the sorted
    // copy pass under the previous tile's steps); they wait in registers
    float4 qn[TILE / BLOCK];
    auto gather = [&](int kt) {
#pragma unroll
      for (int u = 0; u < TILE / BLOCK; ++u) {
        const int k = kt + u * BLOCK + (int)threadIdx.x;
        qn[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < k1) {
          int a = 0, b = rows;  // the segment that holds candidate k: the last r with seg_pref[r] <= k
          while (b - a > 1) {
            const int m = (a + b) >> 1;
            if (seg_pref[m] <= k) a = m;
            else b = m;
          }
          qn[u] = sorted[seg_start[a] + (k - seg_pref[a])];
        }
      }
    };
    __syncthreads();  // seg_pref complete
    if (k0 < k1) gather(k0);
    for (int kt = k0; kt < k1; kt += TILE) {
      __syncthreads();  // the previous tile is no longer read
      const int lim = min(TILE, k1 - kt);
#pragma unroll
      for (int u = 0; u < TILE / BLOCK; ++u) {
        const int kk = u * BLOCK + threadIdx.x;
        uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = make_uint4(0u, 0u, 0x00007F80u, 0u);  // a row past the end: S = +inf
        const float4 q = qn[u];
        if (kk < lim) {
          const float vx = q.x - cx, vy = q.y - cy, vz = q.z - cz;
          const float q2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
          const float scale = pmax2 + q2;
          const float S = scale < kTinyScale ? -INFINITY : q2 - scale * kTauBf16;
          const unsigned int hxy = pack_bf16(vx, vy), lxy = pack_bf16(vx - bf16_lo(hxy), vy - bf16_hi(hxy));
          const unsigned int hzs = pack_bf16(vz, S), lzs = pack_bf16(vz - bf16_lo(hzs), isinf(S) ? 0.0f : S - bf16_hi(hzs));
          r0 = make_uint4(dup_lo(hxy), dup_lo(lxy), dup_hi(hxy), dup_hi(lxy));
          r1 = make_uint4(dup_lo(hzs), dup_lo(lzs), (hzs >> 16) | (lzs & 0xffff0000u), 0u);
        }
        tile[kk] = q;
        aop[kk] = r0;
        aop[TILE + kk] = r1;
      }
      __syncthreads();
      if (kt + TILE < k1) gather(kt + TILE);
      if (stats) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_fill += t - tk0; tk0 = t; }
      const uint4* __restrict__ arow = aop + half * TILE + col;
      uint4 a0 = arow[0];
      for (int st = 0; st < lim; st += 32) {
        const uint4 a = a0;
        a0 = arow[(st + 32) & (TILE - 1)];
        floatx16 acc[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bop[gi]), zero, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        float tail[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) tail[gi] = __builtin_amdgcn_fmed3f(acc[gi][14], acc[gi][15], minus_inf);
        __builtin_amdgcn_sched_barrier(0);
        // the fold keeps its five partial minima: they say which values to look at when the bound is undercut
        float part[G][5], mn[G];
        bool hit = false;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const floatx16 d = acc[gi];
          part[gi][0] = min3f(d[0], d[1], d[2]);
          part[gi][1] = min3f(d[3], d[4], d[5]);
          part[gi][2] = min3f(d[6], d[7], d[8]);
          part[gi][3] = min3f(d[9], d[10], d[11]);
          part[gi][4] = min3f(d[12], d[13], tail[gi]);
          mn[gi] = min3f(min3f(part[gi][0], part[gi][1], part[gi][2]), part[gi][3], part[gi][4]);
          hit |= mn[gi] <= bound[gi];
        }
        if (!(use_prev & 2) && __ballot(hit)) {  // (bit 1: timing experiments only -- no exact path, wrong results)
#pragma unroll
          for (int gi = 0; gi < G; ++gi) {
            if (!__ballot(mn[gi] <= bound[gi])) continue;
            const floatx16 d = acc[gi];
#pragma unroll
            for (int pt = 0; pt < 5; ++pt) {
              if (!__ballot(part[gi][pt] <= bound[gi])) continue;
#pragma unroll
              for (int v = 3 * pt; v < (pt == 4 ? 16 : 3 * pt + 3); ++v) {
                if (d[v] <= bound[gi]) {
                  const int row = 8 * (v >> 2) + 4 * half + (v & 3);
                  if (st + row < lim) {
                    const float4 t = tile[st + row];
                    const float e = dist2(t.x, t.y, t.z, px[gi], py[gi], pz[gi]);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | __float_as_uint(t.w);
                    if (e <= thr && key < best[gi]) {  // (NaN and inf distances fail the first test)
                      best[gi] = key;
                      bqx[gi] = t.x; bqy[gi] = t.y; bqz[gi] = t.z;
                      bound[gi] = __builtin_fmaf(e + p2[gi], 9.5367431640625e-07f, e - p2[gi]);
                    }
                  }
                }
              }
            }
          }
        }
      }
      if (stats) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_steps += t - tk0; tk0 = t; }
    }
  }

  if (stats && lane == 0 && wave == 0) {
    atomicAdd(&stats[0], n_cand * (unsigned long long)(BLOCK / 64) * 32ull * G);
    atomicAdd(&stats[1], t_pre);
    atomicAdd(&stats[2], t_rows);
    atomicAdd(&stats[3], t_fill);
    atomicAdd(&stats[4], t_steps);
    atomicAdd(&stats[5], 1ull);
    atomicAdd(&stats[6], n_cand);
    atomicMax(&stats[7], t_pre + t_rows + t_fill + t_steps);
  }
  // the two half-waves hold the same sources: merge, write under the source's original index
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const unsigned int ohi = (unsigned int)__shfl_xor((int)(best[gi] >> 32), 32, 64);
    const unsigned int olo = (unsigned int)__shfl_xor((int)(unsigned int)best[gi], 32, 64);
    const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
    const unsigned long long k = other < best[gi] ? other : best[gi];
    if (prev && splits == 1) {  // the winner as a point, for the next sweep (the half that holds it writes)
      const float ox = __shfl_xor(bqx[gi], 32, 64), oy = __shfl_xor(bqy[gi], 32, 64), oz = __shfl_xor(bqz[gi], 32, 64);
      const bool mine = best[gi] <= other;
      if (half == 0 && q0 + gi * 32 + col < n_q) {
        const float none = __builtin_nanf("");
        prev[q0 + gi * 32 + col] = k == kEmptyKey ? make_float4(none, none, none, 0.f)
                                                   : make_float4(mine ? bqx[gi] : ox, mine ? bqy[gi] : oy, mine ? bqz[gi] : oz, __uint_as_float((unsigned int)k));
      }
    }
    if (half == 0 && q0 + gi * 32 + col < n_q && k != kEmptyKey) {
      if (splits > 1) atomicMin(&keys[orig[gi]], k);
      else keys[orig[gi]] = k;
    }
  }
}

}  // namespace

// keys (n_s entries at the ORIGINAL source indices, pre-filled with kEmptyKey): the exact nearest neighbour of T * src among
// the grid's points when it lies within sqrt(thr), key = (d2 bits, original target index); empty otherwise.
// src_morton: launch_morton_order's output (finite points, original index in .w).  tgt: the target in its original order (what
// the seeds index).  seed (nullable): the keys of the previous sweep of the same source over the same target.
// prev (nullable, n_q float4 at the sources' Morton positions; used with ONE split per group only): every sweep leaves the
// neighbours it found there as points (.w = original target index), and with use_prev reads the previous sweep's instead of
// the seed keys.  stats (nullable, 8 x u64, zeroed by the caller): pairs offered to the filter, cycles per phase.
hipError_t launch_nn_tile_search(const float4* src_morton, int n_q, const Xform& T, const float4* sorted, const int* cell_start,
                                 const GridDesc& g, const float4* tgt, int n_t, float thr, const unsigned long long* seed,
                                 float4* prev, bool use_prev, unsigned long long* keys, unsigned long long* stats,
                                 hipStream_t stream) {
  if (n_q <= 0 || n_t <= 0) return hipSuccess;
  static const int splits_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_TILE_SPLITS"); return e ? atoi(e) : 4; }();
  const int per_block = (TS_BLOCK / 64) * 32 * TS_G;
  const int grid_x = (n_q + per_block - 1) / per_block;
  const int splits = splits_env < 1 ? 1 : splits_env > 64 ? 64 : splits_env;
  static const int no_exact = [] {
    if (!ICPGPU_DEV_ENV("ICPGPU_TILE_NO_EXACT")) return 0;
    fprintf(stderr, "[icpgpu] WARNING: ICPGPU_TILE_NO_EXACT is set -- the tile search skips its exact path, its results are WRONG (timing experiment)\n");
    return 2;
  }();
  hipLaunchKernelGGL(nn_tile_kernel, dim3(grid_x, splits), dim3(TS_BLOCK), 0, stream, src_morton, n_q, T, sorted, cell_start, g,
                     tgt, n_t, thr, seed, splits == 1 ? prev : nullptr, ((splits == 1 && prev && use_prev) ? 1 : 0) | no_exact, keys, splits, stats);
  return hipGetLastError();
}

}  // namespace icpgpu
