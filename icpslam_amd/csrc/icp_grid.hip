// icp_grid.hip -- uniform-grid accelerated EXACT nearest neighbour for the ICP correspondence step (row a2 of
// SURVEY.md section 8(a)), plus the fused rejection + covariance reduction (a3 + a4).
//
// Replaces the FLANN kd-tree search PCL performs inside `icp.align()` (reached from
// /root/reference/src/icpslam/icp_odometer.cpp:198 and src/icpslam/octree_mapper.cpp:114).  ICP only keeps
// correspondences closer than setMaxCorrespondenceDistance (icp_odometer.cpp:191, 1.0 m), so the search may stop at
// that radius: target points are counting-sorted into a dense uniform grid once per target cloud, and each source
// point scans the 2x2x2 octant of cells it leans towards, then cubes of growing radius, until its best distance is
// provably smaller than anything outside what was scanned -- never more than the ball that a known target point (the
// octant's winner, a cube's, or the neighbour found in the previous sweep) puts around it.  Same arithmetic contract as
// the brute-force kernel (icp_device.h), same (d2, lowest original index) tie-break, so the keys are bit-identical to
// brute force for every matched point.  Two kernels share the machinery (icp_grid_device.h): nn_quad_kernel (four
// points per wave pass; clouds from 32k points) and nn_wave_kernel (one wave per point; small clouds).
//
// HBM-side layout: `sorted` = float4 {x, y, z, original-index bits}, cells x-fastest so a run of cells along x is ONE
// contiguous range (9 range lookups for a 3x3x3 block); `cell_start` = int32 per cell (+1).
#include <math.h>

#include <cstdlib>
#include <cstring>

#include "icp_env.h"
#include "icp_device.h"
#include "icp_grid_device.h"
#include "icp_kernels.h"

namespace icpgpu {
namespace {

// ---- order-preserving float <-> int encoding for atomic min/max ------------------------------------------------
__device__ __forceinline__ int enc_float(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
inline float dec_float(int e) {
  const int i = e >= 0 ? e : e ^ 0x7FFFFFFF;
  float f;
  std::memcpy(&f, &i, 4);
  return f;
}

__global__ void bbox_init_kernel(int* mm) {
  if (threadIdx.x < 3) mm[threadIdx.x] = 0x7FFFFFFF;
  else if (threadIdx.x < 6) mm[threadIdx.x] = (int)0x80000000;
}

__global__ __launch_bounds__(256) void bbox_kernel(const float4* __restrict__ pts, int n, int* __restrict__ mm) {
  int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    if (finite3(p.x, p.y, p.z)) {
      const int e[3] = {enc_float(p.x), enc_float(p.y), enc_float(p.z)};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = min(lo[a], e[a]);
        hi[a] = max(hi[a], e[a]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = min(lo[a], __shfl_down(lo[a], off, 64));
      hi[a] = max(hi[a], __shfl_down(hi[a], off, 64));
    }
  }
  // one set of atomics per workgroup (same-address atomics cost ~12 ns each on this part: keep them few)
  __shared__ int wmm[4][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      wmm[wave][a] = lo[a];
      wmm[wave][3 + a] = hi[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    int v = wmm[0][threadIdx.x];
    for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? min(v, wmm[w][threadIdx.x]) : max(v, wmm[w][threadIdx.x]);
    if (threadIdx.x < 3) atomicMin(&mm[threadIdx.x], v);
    else atomicMax(&mm[threadIdx.x], v);
  }
}

// counts[0 .. n_counts) and the statistics words zeroed by ONE launch (two hipMemsetAsync calls before: 4 + 9 us of host time and
// three fill kernels in front of every count pass -- HIP API trace of a pipeline scan, scripts/r5/r5_api_trace.sh)
__global__ __launch_bounds__(256) void grid_clear_kernel(int* __restrict__ counts, size_t n_counts, int* __restrict__ stats) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  if (t < (size_t)kGridStatInts) stats[t] = 0;
  // 16-byte stores over the aligned middle (hipMalloc'd: the base is aligned), single words for the tail
  const size_t n4 = n_counts >> 2;
  int4* c4 = reinterpret_cast<int4*>(counts);
  for (size_t i = t; i < n4; i += stride) c4[i] = make_int4(0, 0, 0, 0);
  for (size_t i = (n4 << 2) + t; i < n_counts; i += stride) counts[i] = 0;
}

__global__ __launch_bounds__(256) void grid_count_kernel(const float4* __restrict__ pts, int n, GridDesc g,
                                                         int* __restrict__ cell_of_point, int* __restrict__ rank,
                                                         int* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int c = -1;  // non-finite points and points outside the grid are never binned
  if (finite3(p.x, p.y, p.z)) {
    int cx, cy, cz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz);
    if (cx >= 0 && cx < g.nx && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz) c = cz * g.sz + cy * g.sy + cx;
  }
  cell_of_point[i] = c;
  rank[i] = c >= 0 ? atomicAdd(&counts[c], 1) : 0;
}

// ---- exclusive scan of the per-cell counts (three small kernels) ---------------------------------------------
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total) {
  // 256 threads; returns the exclusive prefix of v, *total = block sum (valid in all threads)
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

constexpr int SCAN_PER_THREAD = kScanItems / 256;  // 16

__global__ __launch_bounds__(256) void scan_sums_kernel(const int* __restrict__ counts, int n, int* __restrict__ block_sums,
                                                        int* __restrict__ stats) {
  const int base = blockIdx.x * kScanItems + threadIdx.x * SCAN_PER_THREAD;
  int s = 0, m = 0;
  unsigned long long sq = 0;  // sum of squared populations: sq / sum = the population a random POINT sees in its cell
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    const int v = (base + k < n) ? counts[base + k] : 0;
    s += v;
    m = max(m, v);
    sq += (unsigned long long)v * (unsigned long long)v;
  }
  int tot;
  (void)block_exclusive_scan_256(s, &tot);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m = max(m, __shfl_down(m, off, 64));
    sq += __shfl_down(sq, off, 64);
  }
  // one set of atomics per WORKGROUP, and none for the empty stretches of the table (most of it): same-address atomics
  // cost ~12 ns each on this part
  __shared__ int wmax[4];
  __shared__ unsigned long long wsq[4];
  if ((threadIdx.x & 63) == 0) {
    wmax[threadIdx.x >> 6] = m;
    wsq[threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_sums[blockIdx.x] = tot;
    if (tot > 0) {
      atomicMax(&stats[1], max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
      atomicAdd(reinterpret_cast<unsigned long long*>(stats + 2), (wsq[0] + wsq[1]) + (wsq[2] + wsq[3]));
      atomicAdd(&stats[4], tot);
    }
  }
}

__global__ __launch_bounds__(1024) void scan_top_kernel(int* __restrict__ block_sums, int nb, int* __restrict__ stats) {
  // one workgroup: each thread owns a contiguous slice of the block sums
  const int per = (nb + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(nb, lo + per);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += block_sums[k];
  // exclusive scan of the 1024 slice sums: 64-lane shuffle scans + one scan of the 16 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  __shared__ int wtot[16];
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < wave) base += wtot[w];
    total += wtot[w];
  }
  if (threadIdx.x == 0) stats[0] = total;  // number of binned points
  int run = base + incl - s;
  for (int k = lo; k < hi; ++k) {
    const int v = block_sums[k];
    block_sums[k] = run;
    run += v;
  }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(int* __restrict__ counts, int n, const int* __restrict__ block_sums,
                                                         const int* __restrict__ stats) {
  const int base = blockIdx.x * kScanItems + threadIdx.x * SCAN_PER_THREAD;
  int v[SCAN_PER_THREAD], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    v[k] = (base + k < n) ? counts[base + k] : 0;
    s += v[k];
  }
  int tot;
  int run = block_exclusive_scan_256(s, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    if (base + k < n) counts[base + k] = run;
    run += v[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[n] = stats[0];  // cell_start[ncells] = total
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const float4* __restrict__ pts, int n,
                                                           const int* __restrict__ cell_of_point,
                                                           const int* __restrict__ rank,
                                                           const int* __restrict__ cell_start,
                                                           const int* __restrict__ orig_index,
                                                           float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = cell_of_point[i];
  if (c < 0) return;
  const float4 p = pts[i];
  // the index a search reports: the point's own, or (a grid over the DISTINCT points of a cloud with repeats) the index
  // of its first occurrence in the full cloud
  sorted[cell_start[c] + rank[i]] = make_float4(p.x, p.y, p.z, __int_as_float(orig_index ? orig_index[i] : i));
}

// ---- correspondence search over the grid --------------------------------------------------------------------------
// One wave owns up to 16 consecutive source points.
//  * Preamble, lane-parallel: lane l works for point l/4 and octant row l%4 -- it loads the point, transforms it, bins it
//    and fetches the (lo, len) range of one of the four cell rows of the 2x2x2 octant the point leans towards.  One
//    coalesced read, ~60 VALU instructions and one memory round trip for all 16 points instead of one each.
//  * Per point (wave-uniform values, pulled out of the preamble lanes with v_readlane): the wave walks the non-empty
//    rows two at a time (two independent coalesced reads in flight), 64 candidates per step, 6 flops + one 64-bit
//    compare per candidate; each lane keeps a (d2, original index) minimum, merged with DPP moves.  If the best
//    distance is <= 63/64 * h/2 it is final (~91 % of the points of a converging scan pair); otherwise cubes of Chebyshev
//    radius 1, 2, 4, ... r_max are searched until the best distance is provably inside the cube.
//  * The fused 17-term accumulation keeps one term per lane; keys are written once per wave, coalesced.
constexpr int WQ_MAX_QPW = 16;  // points per wave (fewer for small clouds so that the chip still fills)

constexpr int WQ_WAVES = 4;
constexpr int WQ_BLOCK = WQ_WAVES * 64;

template <bool WRITE_KEYS, bool FUSE_REDUCE, bool LIST_UNMATCHED, bool PACK_SHORT_ROWS>
__global__ __launch_bounds__(WQ_BLOCK) void nn_wave_kernel(const float4* __restrict__ src, int n_s, int qpw, int xcd_map, Xform T,
                                                           const float4* __restrict__ sorted,
                                                           const int* __restrict__ cell_start, GridDesc g, float accept_thr,
                                                           unsigned long long* __restrict__ keys,
                                                           double* __restrict__ partials, int* __restrict__ unmatched,
                                                           int* __restrict__ unmatched_count) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Point j of this wave is k0 + j * stride: consecutive points go to DIFFERENT waves.  Clouds usually arrive in firing
  // or voxel order, where neighbours in memory are neighbours in space; waves that owned 16 consecutive points of a dense
  // patch (hundreds of candidates each) would run 3x longer than the rest (measured), striding spreads them evenly.
  // Workgroups go round-robin to the 8 XCDs (each with a private L2).  When the source is in cell order, XCD x gets the
  // x-th eighth of it (gridDim.x is a multiple of 8): its L2 then only ever sees that part of the target.
  const int lb = xcd_map ? (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int k0 = lb * WQ_WAVES + wave, stride = gridDim.x * WQ_WAVES;

  // fused reduction: lane t < 17 owns term t = qsel * psel (term order of accumulate_pair)
  double acc = 0.0;
  int qi = -1, pi = -1;  // which component of q / p this lane multiplies (-1 -> 1.0, qi 3 -> d2)
  if (lane >= 1 && lane <= 3) pi = lane - 1;
  else if (lane >= 4 && lane <= 6) qi = lane - 4;
  else if (lane >= 7 && lane <= 15) { qi = (lane - 7) / 3; pi = (lane - 7) % 3; }
  else if (lane == 16) qi = 3;

  // preamble: lane -> (point lane/4, octant row lane%4)
  float lpx = 0.f, lpy = 0.f, lpz = 0.f, lmargin = 0.5f;
  int lcx = 0, lcy = 0, lcz = 0, llo = 0, llen = 0;
  bool lfin = false;
  {
    const int sub = lane >> 2, il = k0 + sub * stride;
    if (sub < qpw && il < n_s) {
      const float4 s = src[il];
      xform_point(T, s.x, s.y, s.z, lpx, lpy, lpz);
      lfin = finite3(lpx, lpy, lpz);
      if (lfin) {
        cell_of(g, lpx, lpy, lpz, lcx, lcy, lcz);
        octant_row(cell_start, g, lpx, lpy, lpz, lcx, lcy, lcz, lane & 3, llo, llen, lmargin);
      }
    }
  }
  const unsigned long long fin_mask = __ballot(lfin), row_mask = __ballot(llen > 0);
  const float lsafe = lmargin * g.h * kGridSafety, lsafe_sq = lsafe * lsafe;  // what the octant can certify, per point
  unsigned long long my_key = kEmptyKey;  // lane q keeps the key of point k0 + q

  for (int qq = 0; qq < qpw; ++qq) {
    const int i = k0 + qq * stride, l0 = qq * 4;
    if (i >= n_s) break;
    LaneBest b{kEmptyKey, 0.f, 0.f, 0.f};
    bool found = false;
    const float px = readlane_f(lpx, l0), py = readlane_f(lpy, l0), pz = readlane_f(lpz, l0);
    if ((fin_mask >> l0) & 1ull) {
      sweep_rows(sorted, llo, llen, row_mask & (0xFull << l0), lane, px, py, pz, b);
      found = merge_lanes(b) && __uint_as_float((unsigned int)(b.key >> 32)) <= readlane_f(lsafe_sq, l0);
      if (!found) {
        const int cx = __builtin_amdgcn_readlane(lcx, l0), cy = __builtin_amdgcn_readlane(lcy, l0),
                  cz = __builtin_amdgcn_readlane(lcz, l0);
        found = grow_cubes<PACK_SHORT_ROWS>(sorted, cell_start, g, px, py, pz, cx, cy, cz, lane, b);
      }
    }
    if constexpr (WRITE_KEYS) {
      if (lane == qq) my_key = found ? b.key : kEmptyKey;
    }
    if constexpr (LIST_UNMATCHED) {
      if (!found && lane == 0) unmatched[atomicAdd(unmatched_count, 1)] = i;
    }
    if constexpr (FUSE_REDUCE) {
      const float d2 = __uint_as_float((unsigned int)(b.key >> 32));
      if (found && d2 <= accept_thr) {  // wave-uniform
        const double a = qi < 0 ? 1.0 : (qi == 0 ? (double)b.qx : qi == 1 ? (double)b.qy : qi == 2 ? (double)b.qz : (double)d2);
        const double c = pi < 0 ? 1.0 : (pi == 0 ? (double)px : pi == 1 ? (double)py : (double)pz);
        acc += a * c;
      }
    }
  }
  if constexpr (WRITE_KEYS) {
    if (lane < qpw && k0 + lane * stride < n_s) keys[k0 + lane * stride] = my_key;
  }
  if constexpr (FUSE_REDUCE) {
    __shared__ double wterm[WQ_WAVES][kReduceTerms];
    if (lane < kReduceTerms) wterm[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < kReduceTerms) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < WQ_WAVES; ++w) v += wterm[w][threadIdx.x];
      partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = v;  // term-major: the final reduction reads rows
    }
  }
}

// ---- the same search, four queries at a time ------------------------------------------------------------------------
// One wave still owns 16 source points and runs the same lane-parallel preamble, but the octant stage works on FOUR
// points at once, 16 lanes each (a DPP row): lanes 16g..16g+15 hold points 4g..4g+3 of the wave after the preamble, so
// in pass b group g takes point 4g+b and everything it needs already sits in its own row.  The four octant rows of a
// point are walked as ONE list (lane s takes entries s, s+16, ...), the per-lane minima are merged inside the row with
// 4+4 v_min_u32_dpp (no readlanes, no scalar tie logic), the winner's coordinates come back through ds_bpermute and the
// 16 product terms of the fused reduction go one to a lane (the count is an integer, the d2 sum shares lane 0).  That
// removes most of the per-point fixed cost of nn_wave_kernel (merge, term selection, scalar row set-up: ~70 of ~150 wave
// instructions).  Points the octant cannot certify fall back to the wave-wide cube search, one at a time, as before.
// (the body: nn_quad_kernel below runs it for one scan pair per launch, nn_quad_batch_kernel for one pair per blockIdx.y;
// n_blocks / bx stand for gridDim.x / blockIdx.x of the single-pair launch, so a pair's points go to the same workgroups, in
// the same order, whichever kernel sweeps it: identical partial sums)
template <bool WRITE_KEYS, bool FUSE_REDUCE, bool LIST_UNMATCHED, bool PACK_SHORT_ROWS>
__device__ __forceinline__ void nn_quad_body(const float4* __restrict__ src, int n_s, int qpw, int xcd_map, const Xform& T,
                                             const float4* __restrict__ sorted, const int* __restrict__ cell_start,
                                             const GridDesc& g, float accept_thr, unsigned long long* __restrict__ keys,
                                             double* __restrict__ partials, int* __restrict__ unmatched,
                                             int* __restrict__ unmatched_count, unsigned int* __restrict__ prev_nn, int use_prev,
                                             int cube_start, unsigned long long* __restrict__ count_candidates, int n_blocks,
                                             int bx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp_base = lane & 48, sub = lane & 15;
  // counting runs (bench.py's useful-flop figure, ICPGPU_COUNT_CANDIDATES): target points this wave evaluates, octant lists
  // + cube rows, without the padding re-reads.  nullptr otherwise: the additions below are scalar and skipped.
  unsigned int n_cand = 0;
  const int lb = xcd_map ? (bx & 7) * (n_blocks >> 3) + (bx >> 3) : bx;
  const int k0 = lb * WQ_WAVES + wave, stride = n_blocks * WQ_WAVES;

  // fused reduction: lane s of a row owns one product term (term order of accumulate_pair; s = 0 takes the d2 sum, the
  // count is kept as an integer)
  double acc = 0.0;
  int cnt = 0;
  int qi = -1, pi = -1;
  if (sub == 0) qi = 3;
  else if (sub <= 3) pi = sub - 1;
  else if (sub <= 6) qi = sub - 4;
  else { qi = (sub - 7) / 3; pi = (sub - 7) % 3; }

  // preamble: lane -> (point lane/4, octant row lane%4), as nn_wave_kernel
  float lpx = 0.f, lpy = 0.f, lpz = 0.f, lmargin = 0.5f;
  int lcx = 0, lcy = 0, lcz = 0, llo = 0, llen = 0;
  bool lfin = false;
  const int il = k0 + (lane >> 2) * stride;
  const bool lvalid = (lane >> 2) < qpw && il < n_s;
  // Temporal coherence: prev_nn[i] holds the POSITION (byte offset into `sorted`) of the neighbour point i found in the
  // previous sweep over the SAME target (kNoPrev: none).  Whatever the transform is now, that point is a target point, so
  // its distance bounds the new neighbour's: only the part of the octant inside that ball is read (near convergence: 1-2
  // cells of the 8).  4 bytes per source point each way (a float4 of coordinates until round 5: 6.4 MB of the sweep's
  // 27.5 MB of HBM traffic at 200k x 200k); the point itself comes from `sorted`, which the search is reading anyway.
  const char* __restrict__ sorted_bytes = reinterpret_cast<const char*>(sorted);
  float lbound = __builtin_nanf("");
  if (lvalid) {
    const float4 s = src[il];
    unsigned int ppos = kNoPrev;
    if (use_prev & 1) ppos = prev_nn[il];
    xform_point(T, s.x, s.y, s.z, lpx, lpy, lpz);
    lfin = finite3(lpx, lpy, lpz);
    if (lfin) {
      cell_of(g, lpx, lpy, lpz, lcx, lcy, lcz);
      if (ppos != kNoPrev) {
        const float4 pq = *reinterpret_cast<const float4*>(sorted_bytes + (size_t)ppos);
        lbound = dist2(pq.x, pq.y, pq.z, lpx, lpy, lpz);  // the expression consider() uses: the same bits when it is met again
      }
      octant_row_in_ball(cell_start, g, lpx, lpy, lpz, lcx, lcy, lcz, lane & 3, ball_cells_sq(g, lbound), llo, llen, lmargin);
    }
  }
  const unsigned long long fin_mask = __ballot(lfin);
  const float lsafe = lmargin * g.h * kGridSafety, lsafe_sq = lsafe * lsafe;  // what the octant can certify, per point
  const int n_pass = (__popcll(__ballot(lvalid)) + 15) >> 4;  // 4 lanes per point, 4 points per pass

  // Octant sizes differ a lot between points (median 135 candidates, 90th percentile 360 at 200k x 200k) and a pass
  // lasts as long as its largest group, so the 16 points are ranked by candidate count (largest first, slots without a
  // point last) and pass b takes ranks 4b..4b+3: four points of similar size.  Modelled on the 200k pair: 5.7 -> 4.0
  // loads per point (the wave-per-point kernel: 3.9).
  int slot_of_rank;  // lane 4r (..4r+3): the slot of the point with rank r
  {
    int cand = llen + __builtin_amdgcn_mov_dpp(llen, 0xB1, 0xF, 0xF, true);
    cand += __builtin_amdgcn_mov_dpp(cand, 0x4E, 0xF, 0xF, true);  // all four rows of the slot
    const unsigned int mine = ((lvalid ? (unsigned int)cand + 1u : 0u) << 4) | (unsigned int)(lane >> 2);  // unique per slot
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) rank += (unsigned int)__builtin_amdgcn_readlane((int)mine, 4 * k) > mine ? 1 : 0;
    slot_of_rank = __builtin_amdgcn_ds_permute((4 * rank + (lane & 3)) << 2, lane >> 2);
  }

  for (int pass = 0; pass < n_pass; ++pass) {
    const int ql = 4 * lane_get_i(slot_of_rank, 4 * (4 * pass + (lane >> 4)));  // preamble lane (row 0) of this group's point
    const int qi_src = k0 + (ql >> 2) * stride;                                 // its index in the source cloud
    const bool valid = (ql >> 2) < qpw && qi_src < n_s;
    const float px = lane_get_f(lpx, ql), py = lane_get_f(lpy, ql), pz = lane_get_f(lpz, ql);
    const bool fin = (fin_mask >> ql) & 1ull;
    const int lo0 = lane_get_i(llo, ql), lo1 = lane_get_i(llo, ql + 1), lo2 = lane_get_i(llo, ql + 2), lo3 = lane_get_i(llo, ql + 3);
    const int n0 = lane_get_i(llen, ql), n1 = lane_get_i(llen, ql + 1), n2 = lane_get_i(llen, ql + 2), n3 = lane_get_i(llen, ql + 3);
    // the four rows as one list of L entries: entry v lives at sorted[v + o_r] for the row r it falls into (all in
    // bytes below, modulo 2^32: the launcher only picks this kernel for targets under 4 GiB).  A group with nothing to
    // search (L = 0) re-reads sorted[0]: any real target point may be offered to an exact minimum.
    const int e0 = n0, e1 = e0 + n1, e2 = e1 + n2, L = e2 + n3;
    const unsigned int eb0 = (unsigned int)e0 << 4, eb1 = (unsigned int)e1 << 4, eb2 = (unsigned int)e2 << 4;
    const unsigned int ob0 = (unsigned int)lo0 << 4, ob1 = (unsigned int)(lo1 - e0) << 4, ob2 = (unsigned int)(lo2 - e1) << 4,
                       ob3 = L ? (unsigned int)(lo3 - e2) << 4 : 0u;
    const unsigned int lastb = (unsigned int)max(L - 1, 0) << 4;
    const int l_max = max(max(__builtin_amdgcn_readlane(L, 0), __builtin_amdgcn_readlane(L, 16)),
                          max(__builtin_amdgcn_readlane(L, 32), __builtin_amdgcn_readlane(L, 48)));
    if (count_candidates)
      n_cand += (unsigned int)(__builtin_amdgcn_readlane(L, 0) + __builtin_amdgcn_readlane(L, 16) + __builtin_amdgcn_readlane(L, 32) +
                               __builtin_amdgcn_readlane(L, 48));
    // per lane: the best (d2, original index) key and where that entry sits (its coordinates are fetched once, after the merge)
    unsigned long long bkey = kEmptyKey;
    unsigned int bpos = 0u;
    unsigned int vb = (unsigned int)sub << 4;
    for (int done = 0; done < l_max; done += 32, vb += 512u) {
      const unsigned int ua = min(vb, lastb), ub = min(vb + 256u, lastb);  // past the end: the last entry again (harmless)
      const unsigned int fa = ua + (ua < eb0 ? ob0 : ua < eb1 ? ob1 : ua < eb2 ? ob2 : ob3);
      const unsigned int fb = ub + (ub < eb0 ? ob0 : ub < eb1 ? ob1 : ub < eb2 ? ob2 : ob3);
      const float4 qa = *reinterpret_cast<const float4*>(sorted_bytes + (size_t)fa);
      const float4 qb = *reinterpret_cast<const float4*>(sorted_bytes + (size_t)fb);
      __builtin_amdgcn_sched_barrier(0);  // both reads in flight before either is used
      const unsigned long long ka = ((unsigned long long)__float_as_uint(dist2(qa.x, qa.y, qa.z, px, py, pz)) << 32) | __float_as_uint(qa.w);
      const unsigned long long kb = ((unsigned long long)__float_as_uint(dist2(qb.x, qb.y, qb.z, px, py, pz)) << 32) | __float_as_uint(qb.w);
      if (ka < bkey) { bkey = ka; bpos = fa; }
      if (kb < bkey) { bkey = kb; bpos = fb; }
    }
    // merge inside the row: distance first, then the lowest original index among the ties
    const unsigned int dbits = (unsigned int)(bkey >> 32), idx = (unsigned int)bkey;
    const unsigned int dmin = row16_min_u32(dbits);
    const unsigned int imin = row16_min_u32(dbits == dmin ? idx : 0xFFFFFFFFu);
    const unsigned long long win = __ballot(dbits == dmin && idx == imin);  // >= 1 lane per row
    const int owner = grp_base + __ffs((unsigned int)(win >> grp_base) & 0xFFFFu) - 1;
    unsigned long long gkey = ((unsigned long long)dmin << 32) | imin;
    unsigned int wpos = (unsigned int)lane_get_i((int)bpos, owner);  // where the winner sits (its coordinates are read after the cubes)
    bool found = fin && __uint_as_float(dmin) <= lane_get_f(lsafe_sq, ql);  // an empty row has dmin = NaN bits: false

    // not certified by the octant: the wave-wide cube search, one point at a time
    unsigned long long need = (use_prev & 2) ? 0ull : __ballot(fin && !found && sub == 0);  // bit 1: timing experiments only (the cube search is skipped: wrong results)
    while (need) {
      const int gl = __ffsll((long long)need) - 1, sl = __builtin_amdgcn_readlane(ql, gl);
      need &= need - 1;
      const float ux = readlane_f(lpx, sl), uy = readlane_f(lpy, sl), uz = readlane_f(lpz, sl);
      const int cx = __builtin_amdgcn_readlane(lcx, sl), cy = __builtin_amdgcn_readlane(lcy, sl),
                cz = __builtin_amdgcn_readlane(lcz, sl);
      // the octant's winner (if any) seeds the search: it bounds the ball the cubes have to cover ...
      LanePos c{((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(gkey >> 32), gl) << 32) |
                    (unsigned int)__builtin_amdgcn_readlane((int)gkey, gl),
                (unsigned int)__builtin_amdgcn_readlane((int)wpos, gl)};
      // ... unless last sweep's neighbour is closer (it may lie outside the octant).  It enters with the highest index:
      // the cubes meet the point itself again, same distance bits, and its real index wins the tie.
      const float pb = readlane_f(lbound, sl);
      if (pb < __builtin_inff() && !(__uint_as_float((unsigned int)(c.key >> 32)) <= pb)) {
        c.key = ((unsigned long long)__float_as_uint(pb) << 32) | 0xFFFFFFFFull;
        c.pos = prev_nn[k0 + (sl >> 2) * stride];  // not yet overwritten: this point's pass is this one
      }
      // First cube radius.  With a previous neighbour the seed's distance D is (nearly) the neighbour's own, and the first
      // cube that can certify anything is the one of radius ceil(D / (63/64 h)): the smaller ones would only be walked to
      // be found insufficient (ball pruning makes the larger one cost what the ball holds) -- measured at 200k x 200k,
      // sweeps 2..10 of an alignment: 637 -> 559 us.  In the first sweep of an alignment the seed is the octant's winner,
      // often far from the neighbour (its ball holds several times the candidates; starting at its radius: 175 -> 221 us):
      // there the start is capped (cube_start >> 8: 1 = the radii grow from 1 as before).
      int rho_first = 1;
      {
        const float seed = __uint_as_float((unsigned int)(c.key >> 32));  // NaN: no seed
        const int cap = (use_prev & 1) ? ((cube_start & 1) ? g.r_max : 1) : min((cube_start >> 8) & 0xFF, g.r_max);
        if (seed < __builtin_inff() && cap > 1)
          rho_first = max(1, min((int)ceilf(__builtin_amdgcn_sqrtf(seed) * g.inv_h * (1.0f / kGridSafety)), cap));
      }
      const bool f2 = grow_cubes<PACK_SHORT_ROWS>(sorted, cell_start, g, ux, uy, uz, cx, cy, cz, lane, c, rho_first,
                                                  count_candidates ? &n_cand : nullptr);
      if (grp_base == gl) {
        gkey = c.key;
        wpos = c.pos;
        found = f2;
      }
    }

    if (prev_nn && sub == 0 && valid) {
      // the best point met, certified or not (beyond the gate it is still a target point, hence a bound for the next
      // sweep -- getFitnessScore's ungated one in particular); kNoPrev when nothing was met at all
      const float bd = __uint_as_float((unsigned int)(gkey >> 32));
      prev_nn[qi_src] = (fin && bd == bd) ? wpos : kNoPrev;
    }
    if constexpr (WRITE_KEYS) {
      if (sub == 0 && valid) keys[qi_src] = found ? gkey : kEmptyKey;
    }
    if constexpr (LIST_UNMATCHED) {
      if (sub == 0 && valid && !found) unmatched[atomicAdd(unmatched_count, 1)] = qi_src;
    }
    if constexpr (FUSE_REDUCE) {
      const float d2 = __uint_as_float((unsigned int)(gkey >> 32));
      if (found && d2 <= accept_thr) {
        const float4 w = *reinterpret_cast<const float4*>(sorted_bytes + (size_t)wpos);  // the winner itself
        const float qx = w.x, qy = w.y, qz = w.z;
        // the lane's two factors are picked as FLOATS (selects on loop-invariant lane masks), then widened: the double-
        // precision ternaries compiled into a ladder of divergent branches (~40 instructions per pass)
        const float af = qi == 0 ? qx : qi == 1 ? qy : qi == 2 ? qz : qi == 3 ? d2 : 1.0f;
        const float cf = pi == 0 ? px : pi == 1 ? py : pi == 2 ? pz : 1.0f;
        acc += (double)af * (double)cf;
        cnt += 1;
      }
    }
  }
  if (count_candidates && lane == 0) atomicAdd(count_candidates, (unsigned long long)n_cand);
  if constexpr (FUSE_REDUCE) {
    __shared__ double wterm[WQ_WAVES * 4][16];
    __shared__ int wcnt[WQ_WAVES * 4];
    wterm[wave * 4 + (lane >> 4)][sub] = acc;
    if (sub == 0) wcnt[wave * 4 + (lane >> 4)] = cnt;
    __syncthreads();
    if (threadIdx.x < kReduceTerms) {
      double v = 0.0;
      if (threadIdx.x == 0) {
        int n = 0;
#pragma unroll
        for (int k = 0; k < WQ_WAVES * 4; ++k) n += wcnt[k];
        v = (double)n;
      } else {
        const int s = threadIdx.x == 16 ? 0 : (int)threadIdx.x;
#pragma unroll
        for (int k = 0; k < WQ_WAVES * 4; ++k) v += wterm[k][s];
      }
      partials[(size_t)threadIdx.x * n_blocks + bx] = v;
    }
  }
}


template <bool WRITE_KEYS, bool FUSE_REDUCE, bool LIST_UNMATCHED, bool PACK_SHORT_ROWS>
__global__ __launch_bounds__(WQ_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void nn_quad_kernel(const float4* __restrict__ src, int n_s, int qpw, int xcd_map, Xform T,
                                                           const float4* __restrict__ sorted,
                                                           const int* __restrict__ cell_start, GridDesc g, float accept_thr,
                                                           unsigned long long* __restrict__ keys,
                                                           double* __restrict__ partials, int* __restrict__ unmatched,
                                                           int* __restrict__ unmatched_count, unsigned int* __restrict__ prev_nn,
                                                           int use_prev, int cube_start,
                                                           unsigned long long* __restrict__ count_candidates) {
  nn_quad_body<WRITE_KEYS, FUSE_REDUCE, LIST_UNMATCHED, PACK_SHORT_ROWS>(src, n_s, qpw, xcd_map, T, sorted, cell_start, g, accept_thr, keys,
                                                                        partials, unmatched, unmatched_count, prev_nn, use_prev,
                                                                        cube_start, count_candidates, (int)gridDim.x, (int)blockIdx.x);
}

// Lock-step sweep of several independent scan pairs (icpgpu_align_batch, BASELINE configs 4 / 5): pair step.slot[blockIdx.y]
// of the table, its transform from the kernel arguments, its own workgroup count (workgroups beyond it leave at once).
// OPEN: the ungated sweep of getFitnessScore (keys + the list of points the grid leaves unmatched) instead of the fused one.
template <bool PACK_SHORT_ROWS, bool OPEN>
__global__ __launch_bounds__(WQ_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void nn_quad_batch_kernel(const BatchPair* __restrict__ pairs, BatchStep step,
                                                                                                       int cube_start) {
  // (Measured and dropped in round 5: XCD k taking the k-th eighth of the pair-major list of work units -- one pair per XCD and L2
  // in a group of eight -- instead of an eighth of every pair: 4-7 % slower at every group size; profiles/r05_batch_groups.txt.)
  const int by = (int)blockIdx.y, bx = (int)blockIdx.x;
  const BatchPair& d = pairs[step.slot[by]];
  const int n_blocks = d.blocks;
  if (bx >= n_blocks) return;
  const Xform T = step.T[by];
  GridDesc g = d.g;
  const int use_prev = (int)((step.use_prev_mask >> by) & 1u);
  if constexpr (OPEN) {
    g.r_max = d.r_max_open;
    nn_quad_body<true, false, true, PACK_SHORT_ROWS>(d.src, d.n_s, d.qpw, d.xcd_map, T, d.sorted, d.cell_start, g, 0.f, d.keys, nullptr,
                                                     d.unmatched, d.unmatched_count, d.prev_nn, use_prev, cube_start, nullptr, n_blocks, bx);
  } else {
    nn_quad_body<false, true, false, PACK_SHORT_ROWS>(d.src, d.n_s, d.qpw, d.xcd_map, T, d.sorted, d.cell_start, g, d.accept_thr, nullptr,
                                                      d.partials, nullptr, nullptr, d.prev_nn, use_prev, cube_start, nullptr, n_blocks, bx);
  }
}

}  // namespace

hipError_t launch_bbox(const float4* pts, int n, int* d_minmax6, hipStream_t stream, bool init) {
  if (init) hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, stream, d_minmax6);
  if (n > 0) {
    int blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(bbox_kernel, dim3(blocks), dim3(256), 0, stream, pts, n, d_minmax6);
  }
  return hipGetLastError();
}

void decode_bbox(const int enc[6], float lo[3], float hi[3]) {
  for (int a = 0; a < 3; ++a) {
    lo[a] = dec_float(enc[a]);
    hi[a] = dec_float(enc[3 + a]);
  }
}

hipError_t launch_grid_count(const float4* pts, int n, const GridDesc& g, int* cell_of_point, int* rank_in_cell,
                             int* counts, int* block_sums, int* d_stats, hipStream_t stream) {
  const int ncells = g.nx * g.ny * g.nz;
  {
    const size_t n_counts = (size_t)ncells + 1;
    const size_t want = (n_counts / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(grid_clear_kernel, dim3(blocks), dim3(256), 0, stream, counts, n_counts, d_stats);
  }
  if (n > 0)
    hipLaunchKernelGGL(grid_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pts, n, g, cell_of_point,
                       rank_in_cell, counts);
  const int nb = (ncells + kScanItems - 1) / kScanItems;
  hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(256), 0, stream, counts, ncells, block_sums, d_stats);
  return hipGetLastError();
}

hipError_t launch_grid_finish(const float4* pts, int n, const GridDesc& g, const int* cell_of_point,
                              const int* rank_in_cell, int* counts, int* block_sums, int* d_stats, const int* orig_index,
                              float4* sorted, hipStream_t stream) {
  const int ncells = g.nx * g.ny * g.nz;
  const int nb = (ncells + kScanItems - 1) / kScanItems;
  hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, stream, block_sums, nb, d_stats);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, stream, counts, ncells, block_sums, d_stats);
  if (n > 0)
    hipLaunchKernelGGL(grid_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, pts, n, cell_of_point,
                       rank_in_cell, counts, orig_index, sorted);
  return hipGetLastError();
}

// nn_quad_kernel takes over from nn_wave_kernel at 32k points (measured: 5k 12.1 vs 9.6 us, 10k 17.2 vs 13.8, 20k-30k equal,
// 50k 25 vs 28, 200k 58 vs 71): below that the chip is filled by giving every wave a single point.
static bool quad_enabled() {
  static const bool v = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_QUAD"); return !e || atoi(e) != 0; }();  // 0: nn_wave_kernel
  return v;
}
// counting runs: a device counter every nn_quad_kernel launch adds its evaluated target points to (grid_count_candidates)
static unsigned long long* g_count_candidates = nullptr;
// timing experiments only: ICPGPU_SKIP_UNCERT (uncertified points are dropped -- wrong results, the octant stage's time alone)
static int quad_debug_bits() {
  static const int v = [] {
    if (!ICPGPU_DEV_ENV("ICPGPU_SKIP_UNCERT")) return 0;
    fprintf(stderr, "[icpgpu] WARNING: ICPGPU_SKIP_UNCERT is set -- uncertified points are DROPPED, every result of this process is WRONG "
                    "(a timing experiment's switch, never a production setting)\n");
    return 2;
  }();
  return v;
}
// First radius of the cube search (see nn_quad_kernel).  Bit 0: sweeps with a previous neighbour start at the seed's radius
// (ICPGPU_CUBE_START=0: at 1); bits 8..15: cap of the start radius in sweeps without one (ICPGPU_CUBE_COLD_CAP, default 1).
static int cube_start_mask() {
  static const int v = [] {
    const char* e = ICPGPU_DEV_ENV("ICPGPU_CUBE_START");
    const char* c = ICPGPU_DEV_ENV("ICPGPU_CUBE_COLD_CAP");
    const int warm = e ? (atoi(e) & 1) : 1, cold = c ? atoi(c) : 1;
    return warm | ((cold < 1 ? 1 : cold > 255 ? 255 : cold) << 8);
  }();
  return v;
}
static bool use_quad(int n_s) { return quad_enabled() && n_s >= 4 * 8192; }

// points per wave: up to 16 for large clouds, fewer when that would leave most of the 256 CUs x 8 waves/SIMD idle
static int queries_per_wave(int n_s) {
  static const int forced = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_QPW"); return e ? atoi(e) : 0; }();  // experiments: 4, 8, 12, 16
  if (forced >= 4 && forced <= WQ_MAX_QPW && use_quad(n_s)) return forced & ~3;
  int q = n_s / 8192;
  if (use_quad(n_s)) q = (q + 3) & ~3;  // four points per pass
  if (q < 1) q = 1;
  if (q > WQ_MAX_QPW) q = WQ_MAX_QPW;
  return q;
}

void grid_count_candidates(unsigned long long* device_counter) { g_count_candidates = device_counter; }

bool grid_search_batchable(int n_s, int flags) { return use_quad(n_s) && !(flags & kGridOver4GiB) && queries_per_wave(n_s) > 0; }
int grid_search_qpw(int n_s) { return queries_per_wave(n_s); }

hipError_t launch_nn_grid_search_batch(const BatchPair* d_pairs, const BatchStep& step, int n_active, int max_blocks, bool pack,
                                       bool open_range, hipStream_t stream) {
  if (n_active <= 0 || max_blocks <= 0) return hipSuccess;
  const dim3 grid((unsigned)max_blocks, (unsigned)n_active), block(WQ_BLOCK);
  if (pack && open_range) hipLaunchKernelGGL((nn_quad_batch_kernel<true, true>), grid, block, 0, stream, d_pairs, step, cube_start_mask());
  else if (pack) hipLaunchKernelGGL((nn_quad_batch_kernel<true, false>), grid, block, 0, stream, d_pairs, step, cube_start_mask());
  else if (open_range) hipLaunchKernelGGL((nn_quad_batch_kernel<false, true>), grid, block, 0, stream, d_pairs, step, cube_start_mask());
  else hipLaunchKernelGGL((nn_quad_batch_kernel<false, false>), grid, block, 0, stream, d_pairs, step, cube_start_mask());
  return hipGetLastError();
}

bool grid_search_keeps_prev(int n_s, int flags) { return use_quad(n_s) && !(flags & kGridOver4GiB); }

int grid_search_blocks(int n_s) {
  const int per_block = WQ_WAVES * queries_per_wave(n_s);
  const int nb = (n_s + per_block - 1) / per_block;
  return (nb + 7) & ~7;  // a multiple of the 8 XCDs
}

hipError_t launch_nn_grid_search(const float4* src, int n_s, int flags, const Xform& T, const float4* sorted,
                                 const int* cell_start, const GridDesc& g, float accept_thr, unsigned long long* keys,
                                 double* partials, int* unmatched, int* unmatched_count, hipStream_t stream,
                                 unsigned int* prev_nn, bool use_prev) {
  if (n_s <= 0) return hipSuccess;
  const int blocks = grid_search_blocks(n_s);
  const int qpw = queries_per_wave(n_s), xm = flags & kGridSrcInCellOrder;
  const bool pack = (flags & kGridPackShortRows) != 0;
  dim3 grid(blocks), block(WQ_BLOCK);
  const bool quad = use_quad(n_s) && !(flags & kGridOver4GiB);
#define ICP_LAUNCH_WQP(K, F, U, P)                                                                                      \
  do {                                                                                                                  \
    if (quad)                                                                                                           \
      hipLaunchKernelGGL((nn_quad_kernel<K, F, U, P>), grid, block, 0, stream, src, n_s, qpw, xm, T, sorted, cell_start, \
                         g, accept_thr, keys, partials, unmatched, unmatched_count, prev_nn,                    \
                         ((prev_nn && use_prev) ? 1 : 0) | quad_debug_bits(), cube_start_mask(), g_count_candidates);     \
    else                                                                                                                \
      hipLaunchKernelGGL((nn_wave_kernel<K, F, U, P>), grid, block, 0, stream, src, n_s, qpw, xm, T, sorted, cell_start, \
                         g, accept_thr, keys, partials, unmatched, unmatched_count);                                    \
  } while (0)
#define ICP_LAUNCH_WQ(K, F, U)              \
  do {                                      \
    if (pack) ICP_LAUNCH_WQP(K, F, U, true); \
    else ICP_LAUNCH_WQP(K, F, U, false);     \
  } while (0)
  const bool k = keys != nullptr, f = partials != nullptr, u = unmatched != nullptr;
  if (k && !f && u) ICP_LAUNCH_WQ(true, false, true);
  else if (k && !f) ICP_LAUNCH_WQ(true, false, false);
  else if (!k && f && !u) ICP_LAUNCH_WQ(false, true, false);
  else if (k && f && !u) ICP_LAUNCH_WQ(true, true, false);
  else return hipErrorInvalidValue;
#undef ICP_LAUNCH_WQP
#undef ICP_LAUNCH_WQ
  return hipGetLastError();
}

}  // namespace icpgpu
