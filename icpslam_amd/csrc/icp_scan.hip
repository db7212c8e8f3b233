// icp_scan.hip -- the two generic primitives the widened rows need, hand-written for gfx950 (until round 3 these were
// rocPRIM calls): an exclusive prefix sum of int32 and a STABLE least-significant-digit radix sort of (int32 key, int32
// value) pairs.
//   * exclusive scan: the map's order-preserving compactions (icp_map.hip: winners of an insert, the nn cloud, its distinct
//     points) and the voxel filter's sort path (icp_voxel.hip) -- three launches: tile sums, one workgroup over the tile
//     sums, apply.  Same structure as the cell-table scan of icp_grid.hip, for arbitrary in / out arrays.
//   * radix sort: the voxel filter's FALLBACK path only (clouds over 2 M points, a voxel bucket that exceeds the group
//     kernel's LDS capacity, int32 index wrap-around): PCL's VoxelGrid sorts (cell index, point index) pairs and adds the
//     members of a cell in input order, so the sort must be stable.  4 bits per pass: tile histogram (digit-major) -> scan
//     -> scatter in which every lane ranks its own items in input order.  Not tuned: it is the rare path.
#include <hip/hip_runtime.h>

#include "icp_kernels.h"

namespace icpgpu {
namespace {

constexpr int SC_TILE = 4096, SC_PER = SC_TILE / 256;

// exclusive prefix of v over the 256 threads of the workgroup; *total = the workgroup's sum
__device__ __forceinline__ int wg_exclusive_scan_256(int v, int* total) {
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

__global__ __launch_bounds__(256) void scan_tile_sums_kernel(const int* __restrict__ in, int n, int* __restrict__ tile_sums) {
  const long long base = (long long)blockIdx.x * SC_TILE + threadIdx.x * SC_PER;
  int s = 0;
#pragma unroll
  for (int k = 0; k < SC_PER; ++k) s += (base + k < n) ? in[base + k] : 0;
  int tot;
  (void)wg_exclusive_scan_256(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the tile sums in place; tile_sums[nt] = grand total
__global__ __launch_bounds__(1024) void scan_tiles_kernel(int* __restrict__ tile_sums, int nt) {
  const int per = (nt + 1023) / 1024;
  const int lo = min(nt, (int)threadIdx.x * per), hi = min(nt, lo + per);
  int s = 0;
  for (int k = lo; k < hi; ++k) s += tile_sums[k];
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
  int run = base + incl - s;
  for (int k = lo; k < hi; ++k) {
    const int v = tile_sums[k];
    tile_sums[k] = run;
    run += v;
  }
  if (threadIdx.x == 0) tile_sums[nt] = total;
}

__global__ __launch_bounds__(256) void scan_apply_tiles_kernel(const int* __restrict__ in, int* __restrict__ out, int n,
                                                               const int* __restrict__ tile_sums) {
  const long long base = (long long)blockIdx.x * SC_TILE + threadIdx.x * SC_PER;
  int v[SC_PER], s = 0;
#pragma unroll
  for (int k = 0; k < SC_PER; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int run = wg_exclusive_scan_256(s, &tot) + tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SC_PER; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

// ---- radix sort ---------------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 8, RS_TILE = 256 * RS_ITEMS, RS_BITS = 4, RS_DIGITS = 1 << RS_BITS;

// digit of key k in the pass that starts at `shift`; the pass that contains bit 31 flips it (signed order, as PCL's sort of
// int indices has it); bits from end_bit on do not take part
__device__ __forceinline__ unsigned int rs_digit(int k, int shift, unsigned int live_mask) {
  const unsigned int u = ((unsigned int)k ^ 0x80000000u) & live_mask;
  return (u >> shift) & (RS_DIGITS - 1);
}

__global__ __launch_bounds__(256) void radix_hist_kernel(const int* __restrict__ keys, int n, int shift, unsigned int live_mask,
                                                         int n_tiles, int* __restrict__ hist) {
  __shared__ int cnt[RS_DIGITS];
  if (threadIdx.x < RS_DIGITS) cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * RS_TILE + threadIdx.x * RS_ITEMS;
#pragma unroll
  for (int k = 0; k < RS_ITEMS; ++k)
    if (base + k < n) atomicAdd(&cnt[rs_digit(keys[base + k], shift, live_mask)], 1);
  __syncthreads();
  if (threadIdx.x < RS_DIGITS) hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];  // digit-major
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(const int* __restrict__ keys_in, const int* __restrict__ vals_in, int n,
                                                            int shift, unsigned int live_mask, int n_tiles,
                                                            const int* __restrict__ offsets, int* __restrict__ keys_out,
                                                            int* __restrict__ vals_out) {
  __shared__ int cnt[RS_DIGITS][256 + 1];  // cnt[d][t]: items of digit d in lane t (then: in the lanes before t)
  __shared__ int goff[RS_DIGITS];
  const int t = threadIdx.x;
  const long long base = (long long)blockIdx.x * RS_TILE + t * RS_ITEMS;
  int k[RS_ITEMS], v[RS_ITEMS];
  unsigned int d[RS_ITEMS];
  unsigned long long mine = 0ull;  // 16 four-bit counters: this lane's items per digit (<= 8 each)
#pragma unroll
  for (int j = 0; j < RS_ITEMS; ++j) {
    const bool live = base + j < n;
    k[j] = live ? keys_in[base + j] : 0;
    v[j] = live ? vals_in[base + j] : 0;
    d[j] = live ? rs_digit(k[j], shift, live_mask) : RS_DIGITS;
    if (live) mine += 1ull << (4 * d[j]);
  }
#pragma unroll
  for (int q = 0; q < RS_DIGITS; ++q) cnt[q][t] = (int)((mine >> (4 * q)) & 15ull);
  if (t < RS_DIGITS) goff[t] = offsets[(size_t)t * n_tiles + blockIdx.x];
  __syncthreads();
  // each wave turns four digit rows into exclusive prefixes over the 256 lanes (in lane order = input order)
  {
    const int lane = t & 63, wave = t >> 6;
    for (int q = wave * 4; q < wave * 4 + 4; ++q) {
      int carry = 0;
      for (int c = 0; c < 4; ++c) {
        const int x = cnt[q][c * 64 + lane];
        int incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int y = __shfl_up(incl, off, 64);
          if (lane >= off) incl += y;
        }
        cnt[q][c * 64 + lane] = carry + incl - x;
        carry += __shfl(incl, 63, 64);
      }
    }
  }
  __syncthreads();
  unsigned long long seen = 0ull;  // how many of this lane's earlier items had each digit
#pragma unroll
  for (int j = 0; j < RS_ITEMS; ++j) {
    if (d[j] == RS_DIGITS) continue;
    const int pos = goff[d[j]] + cnt[d[j]][t] + (int)((seen >> (4 * d[j])) & 15ull);
    seen += 1ull << (4 * d[j]);
    keys_out[pos] = k[j];
    vals_out[pos] = v[j];
  }
}

}  // namespace

size_t exclusive_scan_scratch_ints(int n) { return (size_t)((n > 0 ? n : 1) + SC_TILE - 1) / SC_TILE + 2; }

hipError_t launch_exclusive_scan(const int* in, int* out, int n, int* scratch, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int nt = (n + SC_TILE - 1) / SC_TILE;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(nt), dim3(256), 0, stream, in, n, scratch);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, scratch, nt);
  hipLaunchKernelGGL(scan_apply_tiles_kernel, dim3(nt), dim3(256), 0, stream, in, out, n, scratch);
  return hipGetLastError();
}

size_t radix_sort_scratch_ints(int n) {
  const size_t tiles = (size_t)((n > 0 ? n : 1) + RS_TILE - 1) / RS_TILE;
  return 2 * RS_DIGITS * tiles + exclusive_scan_scratch_ints((int)(RS_DIGITS * tiles)) + 8;
}

// keys[0..n) / vals[0..n) sorted by the key bits [0, end_bit) (bit 31, when it takes part, in signed order) into keys[n..2n) /
// vals[n..2n); both arrays hold 2 n ints (ping-pong).  Stable.
hipError_t launch_radix_sort_pairs(int* keys, int* vals, int n, unsigned int end_bit, int* scratch, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int tiles = (n + RS_TILE - 1) / RS_TILE;
  int* hist = scratch;
  int* offs = scratch + (size_t)RS_DIGITS * tiles;
  int* scan_scratch = offs + (size_t)RS_DIGITS * tiles;
  const unsigned int live_mask = end_bit >= 32 ? 0xFFFFFFFFu : ((1u << end_bit) - 1u);
  int passes = (int)((end_bit + RS_BITS - 1) / RS_BITS);
  if (passes < 1) passes = 1;
  int cur = 0;  // which half holds the current order
  for (int p = 0; p < passes; ++p) {
    const int shift = p * RS_BITS;
    int *ki = keys + (size_t)cur * n, *vi = vals + (size_t)cur * n, *ko = keys + (size_t)(1 - cur) * n, *vo = vals + (size_t)(1 - cur) * n;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(tiles), dim3(256), 0, stream, ki, n, shift, live_mask, tiles, hist);
    hipError_t e = launch_exclusive_scan(hist, offs, RS_DIGITS * tiles, scan_scratch, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(tiles), dim3(256), 0, stream, ki, vi, n, shift, live_mask, tiles, offs, ko, vo);
    cur = 1 - cur;
  }
  if (cur == 0) {  // an even number of passes: the result sits in the first halves
    hipError_t e = hipMemcpyAsync(keys + n, keys, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(vals + n, vals, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

}  // namespace icpgpu
