// icp_brute_bf16.hip -- brute-force exact nearest neighbour, the lower bound on the bf16 matrix path (row a2 of SURVEY.md
// section 8(a); north_star's LDS-tiled brute force, the arithmetic of PCL's CorrespondenceEstimation::determineCorrespondences
// reached from /root/reference/src/icpslam/icp_odometer.cpp:198 and src/icpslam/octree_mapper.cpp:114).
//
// Why a second matrix-core kernel: on gfx950 an f32 MFMA and the ordinary vector instructions of the same SIMD do not run side
// by side (scripts/probes/mfma_coissue.cpp, profiles/r03_mfma_coissue.txt: 4 f32 MFMAs + 32 v_fma_f32 take the SUM of their
// times, in one wave or in two), so icp_brute_mfma.hip's two f32 MFMAs + ~13 vector instructions per 1024 pairs cannot pass
// ~71 % of the f32 peak whatever the schedule.  bf16 MFMAs DO overlap with vector work, and one v_mfma_f32_32x32x16_bf16
// (K = 16, 34 cycles) does what the two f32 MFMAs (K = 2 + 2, 128 cycles) do if the operands are split:
//
//   u = p - c, v = q - c (c: centre of the workgroup's sources);  U = -2u
//   x = xh + xl + r,  xh = bf16(x), xl = bf16(x - xh)  (round to nearest even, v_cvt_pk_bf16_f32; |r| <= 2^-16 |x|)
//   s_ij = sum over the coordinates of (Uh + Ul)(vh + vl) -- four exact bf16 x bf16 products each, 12 of the 16 K slots --
//          + (S_h + S_l) * 1,   S = |v|^2 - tau_j  (2 more slots; 2 unused)
//   tau_j = 2^-13 (P^2 + |v_j|^2),  P = max |u| of the workgroup
//   =>  s_ij <= |q_j - p_i|^2 - |u_i|^2   for every pair (error budget below), exactly the f32 kernel's statement with a wider tau.
//
// Everything else is the f32 kernel's: each lane keeps the best EXACT key per source (contract arithmetic on the original
// coordinates, lowest index among ties) and the bound B = its distance - |u|^2 (+ margin); the 16 values a lane receives from a
// 32 x 32 tile are folded with v_min3 and compared ONCE; only a lane that sees s <= B evaluates those targets exactly.  The
// result is the exact key -- same bits as the VALU kernel and the oracle.  Because tau grows with P^2, the sources come in
// MORTON order of their grid cells (launch_morton_order below: a workgroup's 256 sources sit within 1-4 m of each other instead
// of the 1-40 m of x-fastest cell order) and the centre is the middle of their bounding box.
//
// Error budget of the lower bound, relative to (P^2 + |v|^2):
//   operand splits: |u.v - (uh + ul).(vh + vl)| <= (2^-16 + 2^-16)|u||v| per sum, times 2: 2^-14 |u||v| <= 2^-15 (P^2 + |v|^2)
//   S = S_h + S_l + r, |r| <= 2^-16 |S| <= 2^-16 |v|^2 (+ tau);  the MFMA's own accumulation: MEASURED (round 4,
//   scripts/probes/mfma_accum.cpp, profiles/r04_mfma_accumulation.txt: 8192 MFMAs x 1024 outputs against the exact sums, operands
//   chosen to cancel in the 14 K slots used here, exponent spreads up to 30 binades): the adder aligns to the largest product and
//   drops what lies 2^-21 .. 2^-24 below it -- worst |error| = 2^-21.9 of the sum of the |products| (2^-20.8 of the largest
//   one).  The |products| of a bound add up to <= 2 |u||v| (1 + 2^-7) + |v|^2 + tau <= 2 (P^2 + |v|^2), so the accumulation costs
//   <= 2^-20.9 (P^2 + |v|^2); the budget keeps the 2^-17 it assumed until then (a factor 15 above the measurement); the f32
//   kernel's budget for the rounded differences and |v|^2: 2^-19.
//   Sum < 1.9 x 2^-15; tau = 2^-13 leaves a factor two.  MEASURED, not only argued: ICPGPU_MFMA_CHECK_BOUND=1 evaluates every
//   pair exactly and counts the pairs whose bound exceeds what their own distance allows (must be 0) and the worst
//   (s + tau - truth) / (P^2 + |v|^2) seen (tests/test_gpu_brute_bf16.py and a campaign of 4000 random clouds: 1.8e-5 = 2^-15.8 at worst against tau = 2^-13).
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

constexpr float kTauBf16 = 1.220703125e-04f;   // 2^-13
constexpr float kTinyScale = 1e-18f;           // P^2 + |v|^2 below this: products near the denormal range, no bound claimed

__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// two floats -> two bf16 (round to nearest even) in one dword, `lo` in the low half: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
  const floatx2 f = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ unsigned int dup_lo(unsigned int p) { return __builtin_amdgcn_perm(p, p, 0x01000100u); }
__device__ __forceinline__ unsigned int dup_hi(unsigned int p) { return __builtin_amdgcn_perm(p, p, 0x03020302u); }

// G groups of 32 sources per wave (both half-waves hold the same 32 sources; each sees 16 of a step's 32 targets); TILE
// targets per LDS tile; BLOCK threads per workgroup (the A operands of a tile are built once per workgroup: the more sources
// share them, the less they cost per pair)
template <int G, int TILE, int BLOCK, bool CHECK>
__device__ __forceinline__ void nn_brute_bf16_body(const float4* __restrict__ src_sorted, int n_q, const float4* __restrict__ tgt,
                                                   int n_t, const Xform& T, int tgt_per_split, int splits,
                                                   unsigned long long* __restrict__ keys,
                                                   const unsigned long long* __restrict__ seed, int flags,
                                                   unsigned long long* __restrict__ check) {
  // a target tile in LDS, twice: the raw points (for the exact evaluations) and the MFMA's A operands, ready to use: per
  // target one 16-byte record for the lanes of each half-wave (K slots 0-7: x and y products; 8-15: z products and S)
  __shared__ float4 tile[TILE];
  __shared__ uint4 aop[2 * TILE];
  __shared__ float s_box[BLOCK / 64][6];
  __shared__ float s_pmax[BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int q0 = (blockIdx.x * (BLOCK / 64) + wave) * (32 * G);
  const bool no_exact = flags & 1;

  // sources of this wave: group g, column col (transformed: the contract's p = T * s)
  float px[G], py[G], pz[G];
  int orig[G];
  bool valid[G];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k = q0 + g * 32 + col;
    valid[g] = k < n_q;
    const float4 s = src_sorted[min(k, n_q - 1)];
    xform_point(T, s.x, s.y, s.z, px[g], py[g], pz[g]);
    orig[g] = __float_as_int(s.w);
    lo[0] = fminf(lo[0], px[g]); hi[0] = fmaxf(hi[0], px[g]);
    lo[1] = fminf(lo[1], py[g]); hi[1] = fmaxf(hi[1], py[g]);
    lo[2] = fminf(lo[2], pz[g]); hi[2] = fmaxf(hi[2], pz[g]);
  }
  // the workgroup's centre: the middle of its sources' bounding box (they come in Morton order: close together)
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
  float centre[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = s_box[0][a], h = s_box[0][3 + a];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) {
      l = fminf(l, s_box[w][a]);
      h = fmaxf(h, s_box[w][3 + a]);
    }
    centre[a] = 0.5f * l + 0.5f * h;
  }
  const float cx = centre[0], cy = centre[1], cz = centre[2];
  // B operands (K x N = sources): the lanes of half 0 supply K slots 0-7, those of half 1 slots 8-15 (two bf16 per dword):
  //   half 0: (Uxh, Uxl) (Uxh, Uxl) (Uyh, Uyl) (Uyh, Uyl)      half 1: (Uzh, Uzl) (Uzh, Uzl) (1, 1) (0, 0)
  uint4 bop[G];
  float p2[G], pmax2 = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float ux = px[g] - cx, uy = py[g] - cy, uz = pz[g] - cz;
    p2[g] = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    pmax2 = fmaxf(pmax2, p2[g]);
    const float Ux = -2.0f * ux, Uy = -2.0f * uy, Uz = -2.0f * uz;
    const unsigned int hxy = pack_bf16(Ux, Uy), lxy = pack_bf16(Ux - bf16_lo(hxy), Uy - bf16_hi(hxy));
    const unsigned int hz = pack_bf16(Uz, 1.0f), lz = pack_bf16(Uz - bf16_lo(hz), 0.0f);
    const unsigned int dx = (hxy & 0xffffu) | (lxy << 16), dy = (hxy >> 16) | (lxy & 0xffff0000u);
    const unsigned int dz = (hz & 0xffffu) | (lz << 16);
    bop[g] = half ? make_uint4(dz, dz, 0x3F803F80u, 0u) : make_uint4(dx, dx, dy, dy);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pmax2 = fmaxf(pmax2, __shfl_xor(pmax2, off, 64));
  if (lane == 0) s_pmax[wave] = pmax2;
  __syncthreads();
  pmax2 = s_pmax[0];  // P^2 of the WORKGROUP: tau is shared by its waves
#pragma unroll
  for (int w = 1; w < BLOCK / 64; ++w) pmax2 = fmaxf(pmax2, s_pmax[w]);

  unsigned long long best[G];
  float bound[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    best[g] = kEmptyKey;
    bound[g] = INFINITY;
    // Seed (optional): the neighbour this source found in the previous sweep over the same target -- its exact key is a valid
    // candidate and its distance a valid bound from the first tile on.
    if (seed) {
      const unsigned int j = (unsigned int)seed[orig[g]];
      if (j < (unsigned int)n_t) {
        const float4 t = tgt[j];
        const float e = dist2(t.x, t.y, t.z, px[g], py[g], pz[g]);
        if (e < INFINITY) {
          best[g] = ((unsigned long long)__float_as_uint(e) << 32) | j;
          bound[g] = __builtin_fmaf(e + p2[g], 9.5367431640625e-07f, e - p2[g]);
        }
      }
    }
  }

  const int j0 = blockIdx.y * tgt_per_split;
  const int j1 = min(n_t, j0 + tgt_per_split);
  unsigned long long n_violations = 0;
  float worst_ratio = 0.f;
  // target tiles: the NEXT tile's global loads are in flight (in registers) while this one is being worked on
  float4 nxt[TILE / BLOCK];
#pragma unroll
  for (int u = 0; u < TILE / BLOCK; ++u) nxt[u] = tgt[min(j0 + u * BLOCK + (int)threadIdx.x, n_t - 1)];
  for (int jt = j0; jt < j1; jt += TILE) {
    __syncthreads();
    const bool whole = jt + TILE <= j1;  // (uniform) only a split's last tile has rows past the end
#pragma unroll
    for (int u = 0; u < TILE / BLOCK; ++u) {
      const int k = u * BLOCK + threadIdx.x;
      const float4 q = nxt[u];
      tile[k] = q;
      // A operands (M x K = targets): (vxh, vxh) (vxl, vxl) (vyh, vyh) (vyl, vyl) | (vzh, vzh) (vzl, vzl) (Sh, Sl) (0, 0)
      const float vx = q.x - cx, vy = q.y - cy, vz = q.z - cz;
      const float q2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
      const float scale = pmax2 + q2;
      // (scale tiny: the products sit near the denormal range, where the matrix path may flush -- such a target is simply
      // always evaluated exactly; a NaN scale keeps its NaN)
      const float S = scale < kTinyScale ? -INFINITY : q2 - scale * kTauBf16;
      const unsigned int hxy = pack_bf16(vx, vy), lxy = pack_bf16(vx - bf16_lo(hxy), vy - bf16_hi(hxy));
      const unsigned int hzs = pack_bf16(vz, S), lzs = pack_bf16(vz - bf16_lo(hzs), isinf(S) ? 0.0f : S - bf16_hi(hzs));
      uint4 r0 = make_uint4(dup_lo(hxy), dup_lo(lxy), dup_hi(hxy), dup_hi(lxy));
      uint4 r1 = make_uint4(dup_lo(hzs), dup_lo(lzs), (hzs >> 16) | (lzs & 0xffff0000u), 0u);
      if (!whole && jt + k >= j1) {  // a row past the end can never undercut a bound: S = +inf
        r0 = make_uint4(0u, 0u, 0u, 0u);
        r1 = make_uint4(0u, 0u, 0x00007F80u, 0u);
      }
      aop[k] = r0;
      aop[TILE + k] = r1;
    }
    __syncthreads();
    if (jt + TILE < j1) {
#pragma unroll
      for (int u = 0; u < TILE / BLOCK; ++u) nxt[u] = tgt[min(jt + TILE + u * BLOCK + (int)threadIdx.x, n_t - 1)];
    }
    const int lim = min(TILE, j1 - jt);
    const uint4* __restrict__ arow = aop + half * TILE + col;
    float minus_inf;  // (opaque to the compiler: a literal -inf makes it rewrite the v_med3 below as a canonicalising minimum)
    asm volatile("v_mov_b32 %0, 0xff800000" : "=v"(minus_inf));

    // One step = 32 targets against the wave's 32 G sources: G MFMAs, then the fold of their 16 G values per lane.
    auto issue = [&](floatx16 (&acc)[G], const uint4& a) {  // a: this lane's A[m = col][k = 8 half .. 8 half + 7] of the step
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bop[g]), zero, 0, 0, 0);
      }
    };
    auto fold = [&](floatx16 (&acc)[G], int st) {
      // The folds are v_min3_f32 by hand (fminf on MFMA outputs makes the compiler canonicalise every operand first).  Inline
      // assembly is opaque to the compiler's hazard recogniser, which is what inserts the wait states between an MFMA and the
      // first vector read of its result -- so every accumulator is first read by an instruction the compiler CAN see (it
      // waits there), and nothing may be scheduled across.  That instruction does a fold's work too: v_med3_f32(a, b, -inf)
      // = min(a, b); with a NaN among its operands the hardware returns their min3 = -inf, i.e. the step of a non-finite
      // target goes through the exact path (where its NaN / inf distance never wins) instead of being dropped -- correct, and
      // rare.
      __builtin_amdgcn_sched_barrier(0);
      float tail[G];
#pragma unroll
      for (int g = 0; g < G; ++g) tail[g] = __builtin_amdgcn_fmed3f(acc[g][14], acc[g][15], minus_inf);
      __builtin_amdgcn_sched_barrier(0);
      // d[v]: target row 8 * (v / 4) + 4 * half + (v % 4) of this step, source column col.  A NaN (a non-finite point) is
      // dropped by the minimum and fails the comparison below.
      float mn[G];
      bool hit = false;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const floatx16 d = acc[g];
        mn[g] = min3f(min3f(min3f(d[0], d[1], d[2]), min3f(d[3], d[4], d[5]), min3f(d[6], d[7], d[8])),
                      min3f(d[9], d[10], d[11]), min3f(d[12], d[13], tail[g]));
        hit |= mn[g] <= bound[g];
      }
      if (CHECK) {  // (test mode: its own instantiation) every pair exactly: does its bound allow what its own distance would need?
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const floatx16 d = acc[g];
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = 8 * (v >> 2) + 4 * half + (v & 3);
            if (st + row < lim && valid[g]) {
              const float4 t = tile[st + row];
              const float e = dist2(t.x, t.y, t.z, px[g], py[g], pz[g]);
              if (e < INFINITY) {
                if (d[v] > __builtin_fmaf(e + p2[g], 9.5367431640625e-07f, e - p2[g])) ++n_violations;
                const float vx = t.x - cx, vy = t.y - cy, vz = t.z - cz;
                const float scale = pmax2 + __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
                if (scale >= kTinyScale) {
                  // how far the value WITHOUT tau lies above the truth, in units of (P^2 + |v|^2): must stay well below tau
                  const double over = ((double)d[v] + (double)scale * (double)kTauBf16) - ((double)e - (double)p2[g]);
                  worst_ratio = fmaxf(worst_ratio, (float)(over / (double)scale));
                }
              }
            }
          }
        }
      }
      if (!no_exact && __ballot(hit)) {  // rare: some lane's bound is undercut -- those targets are evaluated exactly
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (!__ballot(mn[g] <= bound[g])) continue;
          const floatx16 d = acc[g];
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            if (d[v] <= bound[g]) {
              const int row = 8 * (v >> 2) + 4 * half + (v & 3);
              if (st + row < lim) {
                const float4 t = tile[st + row];
                const float e = dist2(t.x, t.y, t.z, px[g], py[g], pz[g]);
                const unsigned long long key = ((unsigned long long)__float_as_uint(e) << 32) | (unsigned int)(jt + st + row);
                if (e < INFINITY && key < best[g]) {  // NaN and inf distances never win (the VALU kernel's rule)
                  best[g] = key;
                  // everything that can still beat or tie it has  |q - p|^2 - |u|^2 <= e - |u|^2: the bound, widened for the
                  // rounding of p2 and of this subtraction
                  bound[g] = __builtin_fmaf(e + p2[g], 9.5367431640625e-07f, e - p2[g]);
                }
              }
            }
          }
        }
      }
    };
    // two steps per trip; the A operands of a step are read from LDS one step ahead (the read's latency passes under the
    // fold before it instead of in front of the MFMAs); rows up to the end of the tile always hold valid records
    uint4 a0 = arow[0];
    for (int st = 0; st < lim; st += 64) {
      floatx16 acc[G];
      const uint4 a1 = arow[st + 32];
      issue(acc, a0);
      fold(acc, st);
      a0 = arow[(st + 64) & (TILE - 1)];
      if (st + 32 < lim) {
        issue(acc, a1);
        fold(acc, st + 32);
      }
    }
  }

  if (CHECK) {
    if (n_violations) atomicAdd(&check[0], n_violations);
    if (worst_ratio > 0.f) atomicMax(reinterpret_cast<unsigned int*>(&check[1]), __float_as_uint(worst_ratio));
  }
  // the two half-waves hold the same sources: merge, write under the source's original index
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const unsigned int ohi = (unsigned int)__shfl_xor((int)(best[g] >> 32), 32, 64);
    const unsigned int olo = (unsigned int)__shfl_xor((int)(unsigned int)best[g], 32, 64);
    const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
    const unsigned long long k = other < best[g] ? other : best[g];
    if (half == 0 && valid[g]) {
      if (splits > 1) atomicMin(&keys[orig[g]], k);
      else keys[orig[g]] = k;
    }
  }
}

template <int G, int TILE, int BLOCK, bool CHECK = false>
__global__ __launch_bounds__(BLOCK) void nn_brute_bf16_kernel(const float4* __restrict__ src_sorted, int n_q,
                                                                 const float4* __restrict__ tgt, int n_t, Xform T,
                                                                 int tgt_per_split, int splits,
                                                                 unsigned long long* __restrict__ keys,
                                                                 const unsigned long long* __restrict__ seed, int flags,
                                                                 unsigned long long* __restrict__ check) {
  nn_brute_bf16_body<G, TILE, BLOCK, CHECK>(src_sorted, n_q, tgt, n_t, T, tgt_per_split, splits, keys, seed, flags, check);
}

// ---- Morton order of a grid's sorted copy ----------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int spread10(unsigned int x) {  // 10 bits -> every third bit
  x &= 0x3ffu;
  x = (x | (x << 16)) & 0x030000ffu;
  x = (x | (x << 8)) & 0x0300f00fu;
  x = (x | (x << 4)) & 0x030c30c3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ __launch_bounds__(256) void morton_keys_kernel(const float4* __restrict__ sorted, int n, GridDesc g, int shift,
                                                          int* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = sorted[i];
  int cx, cy, cz;
  cell_of(g, p.x, p.y, p.z, cx, cy, cz);
  cx = min(max(cx, 0), g.nx - 1) >> shift;
  cy = min(max(cy, 0), g.ny - 1) >> shift;
  cz = min(max(cz, 0), g.nz - 1) >> shift;
  keys[i] = (int)(spread10((unsigned int)cx) | (spread10((unsigned int)cy) << 1) | (spread10((unsigned int)cz) << 2));
  vals[i] = i;
}

__global__ __launch_bounds__(256) void morton_gather_kernel(const float4* __restrict__ sorted, const int* __restrict__ order, int n,
                                                            float4* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = sorted[order[i]];
}

}  // namespace

size_t morton_order_work_ints(int n) { return 4 * (size_t)n + radix_sort_scratch_ints(n); }

// out[0..n) = sorted[0..n) (a grid's cell-ordered copy: finite points, original index in .w) reordered by the Morton code of
// their cells (cells merged 2^shift to a side until every axis fits 10 bits); stable: ties keep the cell order
hipError_t launch_morton_order(const float4* sorted, int n, const GridDesc& g, int* work, float4* out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int shift = 0;
  while (((g.nx - 1) >> shift) > 1023 || ((g.ny - 1) >> shift) > 1023 || ((g.nz - 1) >> shift) > 1023) ++shift;
  int* keys = work;               // 2 n: input half, output half
  int* vals = work + 2 * (size_t)n;
  int* scratch = work + 4 * (size_t)n;
  hipLaunchKernelGGL(morton_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sorted, n, g, shift, keys, vals);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = launch_radix_sort_pairs(keys, vals, n, 30, scratch, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(morton_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sorted, vals + n, n, out);
  return hipGetLastError();
}

// src_sorted: the source in MORTON order with the ORIGINAL index in .w (launch_morton_order); keys are written at the original
// indices and must be pre-filled with kEmptyKey (points missing from src_sorted -- the non-finite ones -- stay unmatched, and
// target splits merge by atomic min).  seed (optional, n_s keys indexed like `keys`): every source's neighbour from an earlier
// sweep over the SAME target array.  check (optional: 2 x u64, zeroed by the caller): ICPGPU_MFMA_CHECK_BOUND's counters.
hipError_t launch_nn_brute_bf16(const float4* src_sorted, int n_q, const float4* tgt, int n_t, const Xform& T, int num_cus,
                                unsigned long long* keys, const unsigned long long* seed, unsigned long long* check,
                                hipStream_t stream) {
  if (n_q <= 0 || n_t <= 0) return hipSuccess;
  // experiments: ICPGPU_BF16_G (sources per wave / 32: 2 or 4), ICPGPU_BF16_TILE (512 / 1024), ICPGPU_BF16_BLOCK (256 / 512),
  // ICPGPU_MFMA_WAVES (target waves per SIMD the splits aim at), ICPGPU_MFMA_NO_EXACT (timing only: results are wrong)
  static const int g_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BF16_G"); return e ? atoi(e) : 2; }();
  static const int tile_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BF16_TILE"); return e ? atoi(e) : 1024; }();
  static const int block_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BF16_BLOCK"); return e ? atoi(e) : 512; }();
  static const int waves_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_MFMA_WAVES"); return e ? atoi(e) : 32; }();
  static const int no_exact = [] {
    if (!ICPGPU_DEV_ENV("ICPGPU_MFMA_NO_EXACT")) return 0;
    fprintf(stderr, "[icpgpu] WARNING: ICPGPU_MFMA_NO_EXACT is set -- the matrix-core search skips its exact path, every brute-force result "
                    "of this process is WRONG (a timing experiment's switch, never a production setting)\n");
    return 1;
  }();
  const int flags = no_exact;
  const int G = check ? 2 : g_env == 4 ? 4 : 2;  // (the test mode: its own instantiation, one shape)
  const int TILE = check ? 512 : tile_env == 1024 ? 1024 : 512;
  const int BLOCK = check ? 256 : block_env == 512 ? 512 : 256;
  const int per_block = (BLOCK / 64) * 32 * G;
  const int grid_x = (n_q + per_block - 1) / per_block;
  int splits = (num_cus * waves_env * 4 / (BLOCK / 64) + grid_x - 1) / grid_x;
  const int max_splits = (n_t + TILE - 1) / TILE;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int per = (n_t + splits - 1) / splits;
  per = ((per + TILE - 1) / TILE) * TILE;
  splits = (n_t + per - 1) / per;
#define ICPGPU_BF16_LAUNCH(BLK, ...)                                                                                       \
  hipLaunchKernelGGL((__VA_ARGS__), dim3(grid_x, splits), dim3(BLK), 0, stream, src_sorted, n_q, tgt, n_t, T, per, splits, \
                     keys, seed, flags, check)
  if (check) {
    ICPGPU_BF16_LAUNCH(256, nn_brute_bf16_kernel<2, 512, 256, true>);
    return hipGetLastError();
  }
  const int sel = (G == 4 ? 4 : 0) | (TILE == 1024 ? 2 : 0) | (BLOCK == 512 ? 1 : 0);
  switch (sel) {
    case 0: ICPGPU_BF16_LAUNCH(256, nn_brute_bf16_kernel<2, 512, 256>); break;
    case 1: ICPGPU_BF16_LAUNCH(512, nn_brute_bf16_kernel<2, 512, 512>); break;
    case 2: ICPGPU_BF16_LAUNCH(256, nn_brute_bf16_kernel<2, 1024, 256>); break;
    case 3: ICPGPU_BF16_LAUNCH(512, nn_brute_bf16_kernel<2, 1024, 512>); break;
    case 4: ICPGPU_BF16_LAUNCH(256, nn_brute_bf16_kernel<4, 512, 256>); break;
    case 5: ICPGPU_BF16_LAUNCH(512, nn_brute_bf16_kernel<4, 512, 512>); break;
    case 6: ICPGPU_BF16_LAUNCH(256, nn_brute_bf16_kernel<4, 1024, 256>); break;
    default: ICPGPU_BF16_LAUNCH(512, nn_brute_bf16_kernel<4, 1024, 512>); break;
  }
#undef ICPGPU_BF16_LAUNCH
  return hipGetLastError();
}

}  // namespace icpgpu
