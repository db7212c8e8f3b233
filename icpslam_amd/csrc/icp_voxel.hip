// icp_voxel.hip -- voxel-grid down-sampling on the device: SURVEY.md section 8(f2), the step right before the ICP path.
//
// Replaces pcl::VoxelGrid<PointXYZ>::filter as the reference calls it in IcpOdometer::voxelFilterCloud
// (/root/reference/src/icpslam/icp_odometer.cpp:96-101, leaf 0.2 m in /root/reference/config/icpslam.yaml:14):
// cell index = (floor(x/L) - min_b) . (1, div_x, div_x*div_y); one output point per occupied cell = arithmetic mean of
// its points, added in INPUT order (float32: the order is part of the result); output ordered by ascending cell index.
// Two paths, same float32 arithmetic as oracle/icp_oracle.c::orc_voxel_grid, output bit-identical to it:
//   * the direct path (round 2, below): one bucket-distribution pass + a register-resident bitonic sort of LDS-sized groups --
//     three hand-written launches, 37 us for a raw 200k-point scan;
//   * the sort path (round 1; its stable radix sort was a library call until round 3, now icp_scan.hip): 31-bit keys (32 when the index wraps, see
//     launch_voxel_grid), boundary flags + scan, one gather pass -- 14 launches, 105 us; kept for what the direct path hands
//     over (a voxel bucket beyond its LDS capacity, wrapped indices, clouds over 2M points).
#include <hip/hip_runtime.h>

#include "icp_env.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>


#include "icp_kernels.h"

namespace icpgpu {
namespace {

constexpr int kVoxelSentinel = 0x7FFFFFFF;  // key of points that are not binned (non-finite)

__global__ __launch_bounds__(256) void voxel_key_kernel(const float4* __restrict__ pts, int n, float inv_leaf, int minb_x,
                                                        int minb_y, int minb_z, int mul_y, int mul_z,
                                                        int* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int key = kVoxelSentinel;
  if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    const int ix = (int)floorf(p.x * inv_leaf) - minb_x;
    const int iy = (int)floorf(p.y * inv_leaf) - minb_y;
    const int iz = (int)floorf(p.z * inv_leaf) - minb_z;
    key = ix + iy * mul_y + iz * mul_z;
  }
  keys[i] = key;
  vals[i] = i;
}

__global__ __launch_bounds__(256) void voxel_flag_kernel(const int* __restrict__ keys, int n, int* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = keys[i];
  flags[i] = (k != kVoxelSentinel && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// one lane per occupied cell: float32 sum of the cell's points in input order, then the mean
constexpr int VC_BLOCK = 1024;

__global__ __launch_bounds__(VC_BLOCK) void voxel_centroid_kernel(const float4* __restrict__ pts, const int* __restrict__ keys,
                                                                  const int* __restrict__ vals, const int* __restrict__ flags,
                                                                  const int* __restrict__ slots, int n,
                                                                  float4* __restrict__ out) {
  // The sums must run in point order (PCL adds the points of a voxel in input order, in float), but the LOADS need not:
  // every lane gathers the point at its own sorted position into LDS (one parallel round of dependent loads: index, then
  // point), then the first lane of each voxel adds its members from LDS.  Members beyond this workgroup's 1024 positions
  // (rare: the densest 0.2 m voxels of a raw scan hold ~150 points) are fetched eight at a time.
  // Was: key, index and point re-read one member after the other -- 27 memory round trips in a row per 9-point voxel,
  // several hundred for the densest one, which set the kernel's duration (90 us for 200k points).
  __shared__ float4 sp[VC_BLOCK];
  __shared__ int sk[VC_BLOCK];
  const int base = blockIdx.x * VC_BLOCK, i = base + threadIdx.x;
  const int block_end = min(n, base + VC_BLOCK);
  int k = 0;
  if (i < n) {
    k = keys[i];
    sk[threadIdx.x] = k;
    sp[threadIdx.x] = pts[vals[i]];
  }
  __syncthreads();
  if (i >= n || !flags[i]) return;
  float ax = 0.f, ay = 0.f, az = 0.f;
  int j = i;
  for (; j < block_end && sk[j - base] == k; ++j) {
    const float4 p = sp[j - base];
    ax += p.x;
    ay += p.y;
    az += p.z;
  }
  for (bool more = j == block_end && j < n; more;) {
    int kk[8], vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = min(j + u, n - 1);
      kk[u] = keys[idx];
      vv[u] = vals[idx];
    }
    float4 pp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pp[u] = pts[vv[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (more && j < n && kk[u] == k) {
        ax += pp[u].x;
        ay += pp[u].y;
        az += pp[u].z;
        ++j;
      } else {
        more = false;
      }
    }
  }
  const float cnt = (float)(j - i);
  out[slots[i]] = make_float4(ax / cnt, ay / cnt, az / cnt, 1.0f);
}


// ---- round 2: the filter without the library sort -------------------------------------------------------------------
// VERDICT round 1, item 9.  The radix sort (7-8 launches of ~7 us each for a 200k scan, whatever the key width) is replaced by
// ONE distribution pass on the high bits of the cell key and a sort in LDS -- three launches:
//   voxel_hist_kernel     cell key of every point (kept), histogram of key / cells-per-bucket over <= 8192 buckets (privatised in LDS:
//                         a raw scan's near-field buckets take thousands of points, global atomics on them serialise)
//   voxel_scatter_kernel  every workgroup scans the histogram for itself (32 KiB, cheaper than a launch), then writes
//                         (key << 32 | index) of its finite points into the buckets' segments, in ARBITRARY order (local rank
//                         from an LDS counter, one global atomic per non-empty (workgroup, bucket))
//   voxel_group_kernel    workgroup g takes the buckets that start in items [512 g, 512 g + 512): bitonic sort of their
//                         <= 3840 composites in LDS -- ascending key, and inside a key ascending input index, which is
//                         the order PCL adds a voxel's points in and the reason a sort is needed at all --, gathers the
//                         points, and adds every voxel's members in that order: one LANE per (voxel, axis), a float
//                         addition chain cannot be split any further; the centroids go straight behind those of the groups
//                         before it (ascending cell order): every group publishes its voxel count, none waits long
// A group that would exceed the LDS capacity (a bucket of > 3328 points: thousands of points in ONE voxel, or a leaf far
// larger than the point spacing) raises `status`; the host then runs the library-sort path above instead.
constexpr int VX_BINS_LOG2 = 13;
constexpr int VX_BINS = 1 << VX_BINS_LOG2;
constexpr int VX_BLOCK = 1024;
constexpr int VX_PPT = 2;  // points per thread in the histogram / scatter passes
constexpr int VX_QUANTUM = 512;
constexpr int VX_CAP = 3840;   // elements of a group (the sort pads to 4096)
constexpr int VX_SORT = 4096;
constexpr int VX_CHUNK = 256;  // elements one wave sorts without workgroup barriers (4 per lane)
constexpr int VX_PER = VX_BINS / VX_BLOCK;  // histogram bins per thread in the scans

// The filter's parameters as the DEVICE derives them from the bounding box (voxel_plan_kernel): with a plan the kernels below read
// them from memory instead of taking them as arguments, and the host does not wait for the box before it queues them (round 6).
struct VoxelPlan {
  int minb[3], mul_y, mul_z;
  unsigned int cpb;
  int nbins;
  int pre;  // 0: the direct path runs; 1: no finite point; 2: PCL's "leaf size too small" (input returned); 3: the index may wrap (sort path)
};

// (inv_leaf = 1 / leaf in float, as PCL's inverse_leaf_size_; the arithmetic is voxel_filter_device's, operation for operation)
// bbox6 is handed on to bbox_keep[0..5] (what the host reads) and left INITIALISED for the next bounding-box pass (its own init
// launch is then not needed: voxel_filter_device).
__global__ void voxel_plan_kernel(int* __restrict__ bbox6, float inv_leaf, VoxelPlan* __restrict__ plan, int* __restrict__ d_n_out,
                                  int* __restrict__ status, int* __restrict__ bbox_keep) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    const int el = bbox6[a], eh = bbox6[3 + a];
    bbox_keep[a] = el;
    bbox_keep[3 + a] = eh;
    bbox6[a] = 0x7FFFFFFF;          // (bbox_init_kernel's values)
    bbox6[3 + a] = (int)0x80000000;
    lo[a] = __int_as_float(el >= 0 ? el : el ^ 0x7FFFFFFF);
    hi[a] = __int_as_float(eh >= 0 ? eh : eh ^ 0x7FFFFFFF);
  }
  VoxelPlan p;
  p.pre = 0;
  p.cpb = 1u;
  p.nbins = 1;
  p.mul_y = p.mul_z = 0;
  p.minb[0] = p.minb[1] = p.minb[2] = 0;
  if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) {
    p.pre = 1;
  } else {
    long long d[3];
    int divb[3];
    for (int a = 0; a < 3; ++a) {
      d[a] = (long long)((hi[a] - lo[a]) * inv_leaf) + 1;
      p.minb[a] = (int)floorf(lo[a] * inv_leaf);
      divb[a] = (int)floorf(hi[a] * inv_leaf) - p.minb[a] + 1;
    }
    const long long ncells = (long long)divb[0] * divb[1] * divb[2];
    if (d[0] * d[1] * d[2] > (long long)INT32_MAX) p.pre = 2;
    else if (ncells > (long long)INT32_MAX) p.pre = 3;
    else {
      p.mul_y = divb[0];
      p.mul_z = divb[0] * divb[1];
      p.cpb = (unsigned int)((ncells + VX_BINS - 1) / VX_BINS);
      p.nbins = (int)((ncells - 1) / p.cpb) + 1;
    }
  }
  *plan = p;
  if (p.pre != 0) {  // nothing below runs: an empty result, the host reads `pre`
    d_n_out[0] = d_n_out[1] = 0;
    *status = 0;
  }
}

__global__ __launch_bounds__(VX_BLOCK) void voxel_hist_kernel(const float4* __restrict__ pts, int n, float inv_leaf, int minb_x,
                                                              int minb_y, int minb_z, int mul_y, int mul_z, unsigned int cpb, int nbins,
                                                              int* __restrict__ keys, int* __restrict__ relpos,
                                                              int* __restrict__ hist, const VoxelPlan* __restrict__ plan) {
  if (plan) {  // (uniform: scalar loads)
    if (plan->pre != 0) return;
    minb_x = plan->minb[0];
    minb_y = plan->minb[1];
    minb_z = plan->minb[2];
    mul_y = plan->mul_y;
    mul_z = plan->mul_z;
    cpb = plan->cpb;
    nbins = plan->nbins;
  }
  __shared__ int lh[VX_BINS];
  for (int b = threadIdx.x; b < nbins; b += VX_BLOCK) lh[b] = 0;
  __syncthreads();
  int key[VX_PPT], rank[VX_PPT];
#pragma unroll
  for (int k = 0; k < VX_PPT; ++k) {
    const int i = (blockIdx.x * VX_PPT + k) * VX_BLOCK + threadIdx.x;
    key[k] = kVoxelSentinel;
    rank[k] = 0;
    if (i < n) {
      const float4 p = pts[i];
      if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        const int ix = (int)floorf(p.x * inv_leaf) - minb_x;
        const int iy = (int)floorf(p.y * inv_leaf) - minb_y;
        const int iz = (int)floorf(p.z * inv_leaf) - minb_z;
        key[k] = ix + iy * mul_y + iz * mul_z;
        rank[k] = atomicAdd(&lh[(unsigned int)key[k] / cpb], 1);  // position among this workgroup's points of the bucket (any order)
      }
      keys[i] = key[k];
    }
  }
  __syncthreads();
  // the workgroup's points of bucket b take the next lh[b] places of the bucket, wherever the other workgroups' stand
  for (int b = threadIdx.x; b < nbins; b += VX_BLOCK) {
    const int c = lh[b];
    if (c) lh[b] = atomicAdd(&hist[b], c);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < VX_PPT; ++k) {
    const int i = (blockIdx.x * VX_PPT + k) * VX_BLOCK + threadIdx.x;
    if (key[k] != kVoxelSentinel) relpos[i] = lh[(unsigned int)key[k] / cpb] + rank[k];
  }
}

// exclusive scan over the workgroup's 1024 threads (one value each); *total = the sum
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* wsum /* 16 ints of LDS */, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < VX_BLOCK / 64; ++w) {
    const int s = wsum[w];
    base += w < wave ? s : 0;
    all += s;
  }
  __syncthreads();  // wsum may be reused
  if (total) *total = all;
  return base + inc - v;
}

__global__ __launch_bounds__(VX_BLOCK) void voxel_scatter_kernel(const int* __restrict__ keys, const int* __restrict__ relpos,
                                                                 int n, unsigned int cpb, int nbins, const int* __restrict__ hist,
                                                                 int n_groups, int4* __restrict__ group_range,
                                                                 int* __restrict__ status,
                                                                 unsigned long long* __restrict__ comp, const VoxelPlan* __restrict__ plan) {
  if (plan) {
    if (plan->pre != 0) return;
    cpb = plan->cpb;
  }
  __shared__ int st[VX_BINS + 1];  // first item of every bucket
  __shared__ int wsum[VX_BLOCK / 64];
  // this thread's eight buckets of the histogram (two 16-byte loads) and its points, all in flight together
  int v[VX_PER];
  {
    const int4* h4 = reinterpret_cast<const int4*>(hist) + threadIdx.x * (VX_PER / 4);
#pragma unroll
    for (int q = 0; q < VX_PER / 4; ++q) {
      const int4 t = h4[q];  // (bins beyond nbins are zero: the buffer is, and nothing counts into them)
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
  int key[VX_PPT], rel[VX_PPT];
#pragma unroll
  for (int k = 0; k < VX_PPT; ++k) {
    const int i = (blockIdx.x * VX_PPT + k) * VX_BLOCK + threadIdx.x;
    key[k] = i < n ? keys[i] : kVoxelSentinel;
    rel[k] = key[k] != kVoxelSentinel ? relpos[i] : 0;
  }
  int s = 0;
#pragma unroll
  for (int k = 0; k < VX_PER; ++k) s += v[k];
  int total = 0;
  int off = block_exclusive_scan_1024(s, wsum, &total);
#pragma unroll
  for (int k = 0; k < VX_PER; ++k) {
    st[threadIdx.x * VX_PER + k] = off;   // (buckets beyond nbins: = total)
    off += v[k];
  }
  if (threadIdx.x == 0) st[VX_BINS] = total;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < VX_PPT; ++k) {
    const int i = (blockIdx.x * VX_PPT + k) * VX_BLOCK + threadIdx.x;
    if (key[k] != kVoxelSentinel)
      comp[st[(unsigned int)key[k] / cpb] + rel[k]] = ((unsigned long long)(unsigned int)key[k] << 32) | (unsigned int)i;
  }
  if (blockIdx.x == 0) {
    // group g = the buckets whose first item lies in [512 g, 512 g + 512): its items are [first start >= 512 g, first
    // start >= 512 (g + 1)) -- lower bounds in the (non-decreasing) table, whose entries from nbins on equal the total
    if (threadIdx.x == 0) *status = 0;
    for (int g = threadIdx.x; g <= n_groups; g += VX_BLOCK) {
      const int want = g * VX_QUANTUM;
      int a = 0, b = VX_BINS;  // answer in [a, b]
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (st[mid] >= want) b = mid; else a = mid + 1;
      }
      const int first = st[a] >= want ? st[a] : total;
      const int first_key = (int)min((unsigned long long)a * cpb, 0x7FFFFFFFull);  // bucket a begins at cell a * cpb
      if (g < n_groups) {
        group_range[g].x = first;
        group_range[g].z = first_key;
      }
      if (g > 0) {
        group_range[g - 1].y = first;
        group_range[g - 1].w = first_key;
      }
    }
  }
}

// ---- the sort: a normalised bitonic network held in REGISTERS -------------------------------------------------------
// Lane l of wave w holds elements 256 w + 4 l .. + 3 of the group.  Every merge of size k starts with a "flip" (element i
// against i ^ (k - 1)) and goes on with strides k/4 .. 1 (i against i ^ j); every compare-exchange leaves the smaller
// value at the lower index, so the +inf padding behind the group's m elements never moves and whatever would touch a
// chunk that is all padding is simply skipped -- the work follows m, not the next power of two.  Strides 1 and 2 stay
// inside a lane, strides 4 .. 128 cross lanes (DPP moves, ds_swizzle, bpermute: no LDS memory), strides >=
// 256 cross waves through LDS: store four elements, barrier, load the partner's four, barrier.  (The first version ran
// every step through LDS -- two round trips of LDS latency per step, 12 us for 1024 elements, 33 us for 4096: 80 % of
// the filter's time.)
typedef unsigned long long vx_u64;
typedef unsigned int vx_u32;
template <typename T>
__device__ __forceinline__ void vx_ce(T& a, T& b) {
  const bool lt = a < b;
  const T lo = lt ? a : b, hi = lt ? b : a;
  a = lo;
  b = hi;
}
// the value lane (l ^ MASK) holds: DPP moves where the pattern has one (1, 2, 3, 4, 7, 8, 15: plain VALU instructions),
// ds_swizzle inside 32 lanes (16, 31), ds_bpermute beyond (32, 63)
template <int MASK>
__device__ __forceinline__ vx_u32 vx_lane_xor(vx_u32 x, int lane) {
  // (mov_dpp, not update_dpp: every lane is written, so there is no "old" value to copy first, and a plain DPP move with
  // full masks is what the compiler folds into the min / max / compare that consumes it)
  const int v = (int)x;
  if constexpr (MASK == 1) return (vx_u32)__builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
  else if constexpr (MASK == 2) return (vx_u32)__builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  else if constexpr (MASK == 3) return (vx_u32)__builtin_amdgcn_mov_dpp(v, 0x1B, 0xF, 0xF, true);   // quad_perm [3,2,1,0]
  else if constexpr (MASK == 4) {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                          // row_shl:4 into banks 0, 2
    return (vx_u32)__builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);                         // row_shr:4 into banks 1, 3
  } else if constexpr (MASK == 7) return (vx_u32)__builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  else if constexpr (MASK == 8) return (vx_u32)__builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
  else if constexpr (MASK == 15) return (vx_u32)__builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true); // row_mirror
  else if constexpr (MASK == 16) return (vx_u32)__builtin_amdgcn_ds_swizzle(v, 0x401F);
  else if constexpr (MASK == 31) return (vx_u32)__builtin_amdgcn_ds_swizzle(v, 0x7C1F);
  else return (vx_u32)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, v);
}
template <int MASK>
__device__ __forceinline__ vx_u64 vx_lane_xor(vx_u64 v, int lane) {
  const vx_u32 lo = vx_lane_xor<MASK>((vx_u32)v, lane);
  const vx_u32 hi = vx_lane_xor<MASK>((vx_u32)(v >> 32), lane);
  return ((vx_u64)hi << 32) | lo;
}
// keep the smaller (low side) or the larger (high side) of mine and the partner's
__device__ __forceinline__ vx_u64 vx_keep(vx_u64 mine, vx_u64 other, bool low) { return ((mine < other) == low) ? mine : other; }
__device__ __forceinline__ vx_u32 vx_keep(vx_u32 mine, vx_u32 other, bool low) {
  const vx_u32 lo = min(mine, other), hi = max(mine, other);  // (v_min_u32 / v_max_u32 take the DPP operand directly)
  return low ? lo : hi;
}

// strides J .. 4 across lanes (J <= 128), then 2 and 1 inside the lane
template <int J, typename T>
__device__ __forceinline__ void vx_strides_in_wave(T v[4], int lane) {
  if constexpr (J >= 4) {
    const bool low = (lane & (J >> 2)) == 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = vx_keep(v[c], vx_lane_xor<(J >> 2)>(v[c], lane), low);
    vx_strides_in_wave<(J >> 1)>(v, lane);
  } else {
    vx_ce(v[0], v[2]);
    vx_ce(v[1], v[3]);
    vx_ce(v[0], v[1]);
    vx_ce(v[2], v[3]);
  }
}
template <int K, typename T>
__device__ __forceinline__ void vx_merge_in_wave(T v[4], int lane) {  // 8 <= K <= 256
  const bool low = (lane & (K >> 3)) == 0;
  T o[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = vx_lane_xor<(K >> 2) - 1>(v[3 - c], lane);
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = vx_keep(v[c], o[c], low);
  vx_strides_in_wave<(K >> 2)>(v, lane);
}
template <typename T>
__device__ __forceinline__ void vx_sort_chunk(T v[4], int lane) {
  vx_ce(v[0], v[1]);
  vx_ce(v[2], v[3]);
  vx_ce(v[0], v[3]);
  vx_ce(v[1], v[2]);
  vx_ce(v[0], v[1]);
  vx_ce(v[2], v[3]);
  vx_merge_in_wave<8>(v, lane);
  vx_merge_in_wave<16>(v, lane);
  vx_merge_in_wave<32>(v, lane);
  vx_merge_in_wave<64>(v, lane);
  vx_merge_in_wave<128>(v, lane);
  vx_merge_in_wave<256>(v, lane);
}
// a lane's four consecutive elements to / from LDS, 16 bytes at a time
__device__ __forceinline__ void vx_store4(vx_u32* p, const vx_u32 v[4]) {
  typedef vx_u32 u32x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u32x4*>(p) = u32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void vx_load4(const vx_u32* p, vx_u32 o[4]) {
  typedef vx_u32 u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 t = *reinterpret_cast<const u32x4*>(p);
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ __forceinline__ void vx_store4(vx_u64* p, const vx_u64 v[4]) {
  typedef vx_u64 u64x2 __attribute__((ext_vector_type(2)));
  reinterpret_cast<u64x2*>(p)[0] = u64x2{v[0], v[1]};
  reinterpret_cast<u64x2*>(p)[1] = u64x2{v[2], v[3]};
}
__device__ __forceinline__ void vx_load4(const vx_u64* p, vx_u64 o[4]) {
  typedef vx_u64 u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 a = reinterpret_cast<const u64x2*>(p)[0], b = reinterpret_cast<const u64x2*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}

// The whole group in ascending order into arr[0 .. m) (LDS); v = the lane's four elements (padding = all ones).
template <typename T>
__device__ __forceinline__ void vx_sort_group(T v[4], T* arr, int m, int m_pad) {
  const int lane = threadIdx.x & 63, chunk = threadIdx.x >> 6, chunk_base = chunk * VX_CHUNK;
  const bool live = chunk_base < m;  // (a chunk of nothing but padding takes no part)
  if (live) vx_sort_chunk(v, lane);
  for (int k = 2 * VX_CHUNK; k <= m_pad; k <<= 1) {
    // the flip across chunks: chunk c against c ^ (k/256 - 1), lane l against 63 - l, the four elements reversed
    {
      const int pc = chunk ^ ((k >> 8) - 1);
      const bool both = live && pc * VX_CHUNK < m;
      if (both) vx_store4(arr + chunk_base + 4 * lane, v);
      __syncthreads();
      if (both) {
        T t[4];
        vx_load4(arr + pc * VX_CHUNK + 4 * (63 - lane), t);
        const bool low = (chunk & (k >> 9)) == 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = vx_keep(v[c], t[3 - c], low);
      }
      __syncthreads();
    }
    for (int j = k >> 2; j >= VX_CHUNK; j >>= 1) {
      const int pc = chunk ^ (j >> 8);
      const bool both = live && pc * VX_CHUNK < m;
      if (both) vx_store4(arr + chunk_base + 4 * lane, v);
      __syncthreads();
      if (both) {
        T t[4];
        vx_load4(arr + pc * VX_CHUNK + 4 * lane, t);
        const bool low = (chunk & (j >> 8)) == 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = vx_keep(v[c], t[c], low);
      }
      __syncthreads();
    }
    if (live) vx_strides_in_wave<128>(v, lane);
  }
  if (live) vx_store4(arr + chunk_base + 4 * lane, v);
  __syncthreads();
}

// number of bits of x (0 for 0)
__device__ __forceinline__ int vx_bits(unsigned int x) { return 32 - __clz(x); }

__global__ __launch_bounds__(VX_BLOCK) void voxel_group_kernel(const float4* __restrict__ pts, int n,
                                                               const unsigned long long* __restrict__ comp,
                                                               const int4* __restrict__ group_range,
                                                               unsigned long long* __restrict__ published,
                                                               unsigned int epoch, float4* __restrict__ out,
                                                               int* __restrict__ d_n_out, int* __restrict__ hist, int nbins,
                                                               int* __restrict__ status, long long* __restrict__ dbg,
                                                               int test_stall, unsigned long long* __restrict__ clear_word,
                                                               const VoxelPlan* __restrict__ plan) {
  // (the accumulator of the kernel that may follow -- publish_cloud_kernel's fingerprint -- starts from zero)
  if (clear_word && blockIdx.x == 0 && threadIdx.x == 0) *clear_word = 0ull;
  if (plan) {
    if (plan->pre != 0) return;
    nbins = plan->nbins;
  }
  long long stamp[6] = {0, 0, 0, 0, 0, 0};
#define VX_STAMP(k) do { if (dbg && threadIdx.x == 0) stamp[k] = (long long)wall_clock64(); } while (0)
  VX_STAMP(0);
  // 45 KiB: the composites while they are sorted (<= 32 KiB), then x / y / z of the gathered points (rows one word apart
  // in bank so that the three axis lanes of a voxel do not collide)
  constexpr int ROW = VX_CAP + 1;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * ROW * 4];
  __shared__ unsigned short head_pos[VX_CAP + 2];  // element index of the voxel with local rank r; [heads] = m
  __shared__ int wsum[VX_BLOCK / 64];
  static_assert(3 * ROW * 4 >= VX_SORT * 8, "the sort array aliases the coordinate rows");
  float* coord = reinterpret_cast<float*>(lds);
  const int g = blockIdx.x, tid = threadIdx.x;
  const int4 rg = group_range[g];  // items [x, y), cell keys [z, w)
  const int gs = rg.x, m = rg.y - rg.x;
  // leave the histogram zero for the next call (nothing reads it any more)
  for (int b = g * VX_BLOCK + tid; b < nbins; b += gridDim.x * VX_BLOCK) hist[b] = 0;
  // Where the group's centroids go: behind those of all the groups before it (ascending cell order).  Every group
  // PUBLISHES its number of voxels as soon as it knows it -- (epoch << 32 | count), the epoch telling this call's word from
  // an older one -- and adds up its predecessors' words once it needs the offset; by then they have been out for
  // microseconds.  Workgroups start in index order and none waits for a later one, so the wait cannot deadlock; 20 ms
  // without an answer gives up (status 2: the host runs the sort path).
  auto publish = [&](int count) {
    if (test_stall && g == 0 && tid == 0) {  // test hook: group 0 answers after everybody's patience has run out
      const long long t0 = (long long)wall_clock64();
      while ((long long)wall_clock64() - t0 < 3000000) __builtin_amdgcn_s_sleep(64);
    }
    if (tid == 0)
      __hip_atomic_store(&published[g], ((unsigned long long)epoch << 32) | (unsigned int)count, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  };
  auto voxels_before = [&]() -> int {
    int part = 0;
    bool ok = true;
    for (int i = tid; i < g; i += VX_BLOCK) {
      unsigned long long w;
      const long long t0 = (long long)wall_clock64();
      while ((unsigned int)((w = __hip_atomic_load(&published[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != epoch) {
        __builtin_amdgcn_s_sleep(4);
        if ((long long)wall_clock64() - t0 > 2000000) {
          ok = false;
          break;
        }
      }
      part += (int)(unsigned int)w;
    }
    if (!ok) *status = 2;
    int total = 0;
    (void)block_exclusive_scan_1024(part, wsum, &total);
    return __syncthreads_or(ok ? 0 : 1) ? -1 : total;  // -1: no offset (whatever the stale words held must not place a write)
  };
  if (m <= 0 || m > VX_CAP) {
    publish(0);
    if (m > VX_CAP && tid == 0) *status = 1;
    if (g == (int)gridDim.x - 1) {  // the last group reports the number of cells
      const int before = voxels_before();
      if (tid == 0) {
        d_n_out[0] = max(before, 0);
        d_n_out[1] = 0;
      }
    }
    return;
  }
  int m_pad = VX_CHUNK;
  while (m_pad < m) m_pad <<= 1;
  // A group of few buckets (all the crowded ones are) sorts 32-bit words -- (key - first key of the group) above the point
  // index -- when the two fit: half the exchanges, single-instruction compares; the others sort (key << 32 | index).
  const int idx_bits = vx_bits((unsigned int)(n - 1)) > 0 ? vx_bits((unsigned int)(n - 1)) : 1;
  const bool narrow = vx_bits((unsigned int)(rg.w - rg.z - 1)) + idx_bits <= 31;
  constexpr int OWN = VX_SORT / VX_BLOCK;
  bool head[OWN];
  unsigned int pidx[OWN];
  {
    vx_u64 c64[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = tid * 4 + c;  // (= chunk_base + 4 lane + c)
      c64[c] = e < m ? comp[gs + e] : ~0ull;
    }
    VX_STAMP(1);
    if (narrow) {
      vx_u32* arr = reinterpret_cast<vx_u32*>(lds);
      vx_u32 v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        v[c] = c64[c] == ~0ull ? ~0u : ((((vx_u32)(c64[c] >> 32) - (vx_u32)rg.z) << idx_bits) | (vx_u32)c64[c]);
      vx_sort_group(v, arr, m, m_pad);
      const vx_u32 mask = (1u << idx_bits) - 1u;
#pragma unroll
      for (int k = 0; k < OWN; ++k) {
        const int e = tid * OWN + k;
        head[k] = false;
        pidx[k] = 0;
        if (e < m) {
          const vx_u32 x = arr[e];
          head[k] = e == 0 || (arr[e - 1] >> idx_bits) != (x >> idx_bits);
          pidx[k] = x & mask;
        }
      }
    } else {
      vx_u64* arr = reinterpret_cast<vx_u64*>(lds);
      vx_sort_group(c64, arr, m, m_pad);
#pragma unroll
      for (int k = 0; k < OWN; ++k) {
        const int e = tid * OWN + k;
        head[k] = false;
        pidx[k] = 0;
        if (e < m) {
          const vx_u64 x = arr[e];
          head[k] = e == 0 || (vx_u32)(arr[e - 1] >> 32) != (vx_u32)(x >> 32);
          pidx[k] = (vx_u32)x;
        }
      }
    }
  }
  VX_STAMP(2);
  // each thread owns four consecutive sorted elements: gather, voxel ranks
  float4 p[OWN];
  int n_heads = 0;
#pragma unroll
  for (int k = 0; k < OWN; ++k) {
    const int e = tid * OWN + k;
    p[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < m) p[k] = pts[pidx[k]];
    n_heads += head[k] ? 1 : 0;
  }
  int heads = 0;
  int rank = block_exclusive_scan_1024(n_heads, wsum, &heads);  // (its barriers: the composites are dead from here on)
  publish(heads);
#pragma unroll
  for (int k = 0; k < OWN; ++k) {
    const int e = tid * OWN + k;
    if (e < m) {
      coord[e] = p[k].x;
      coord[ROW + e] = p[k].y;
      coord[2 * ROW + e] = p[k].z;
      if (head[k]) head_pos[rank++] = (unsigned short)e;
    }
  }
  if (tid == 0) head_pos[heads] = (unsigned short)m;
  const int before = voxels_before();  // (its barriers publish coord / head_pos)
  if (before < 0) {  // gave up waiting (status 2): nothing is written, the host runs the sort path
    if (tid == 0 && g == (int)gridDim.x - 1) d_n_out[0] = d_n_out[1] = 0;
    return;
  }
  VX_STAMP(3);
  // one lane per (voxel, axis): the members in sorted order = input order
  for (int q = tid; q < 3 * heads; q += VX_BLOCK) {
    const int r = q / 3, axis = q - 3 * r;
    const int e = head_pos[r], len = (int)head_pos[r + 1] - e;
    const float* c = coord + axis * ROW + e;
    float s = 0.f + c[0];
    int j = 1;
    // up to the next 16-byte boundary one by one, then eight members per pair of 16-byte LDS reads, the next pair in flight
    // while this one is added (the additions are one dependent chain: 4 cycles each is the floor)
    for (; j < len && ((axis * ROW + e + j) & 3) != 0; ++j) s += c[j];
    if (j + 8 <= len) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      const v4f* c4 = reinterpret_cast<const v4f*>(c + j);
      v4f a = c4[0], b = c4[1];
      int left = (len - j) >> 3;  // full batches
      j += left << 3;
      while (--left > 0) {
        c4 += 2;
        const v4f na = c4[0], nb = c4[1];
        s += a.x; s += a.y; s += a.z; s += a.w;
        s += b.x; s += b.y; s += b.z; s += b.w;
        a = na;
        b = nb;
      }
      s += a.x; s += a.y; s += a.z; s += a.w;
      s += b.x; s += b.y; s += b.z; s += b.w;
    }
    for (; j < len; ++j) s += c[j];
    float* o = reinterpret_cast<float*>(out + before + r);
    o[axis] = s / (float)len;
    if (axis == 0) o[3] = 1.0f;
  }
  if (dbg) __syncthreads();
  VX_STAMP(4);
  if (tid == 0) {
    if (g == (int)gridDim.x - 1) {
      d_n_out[0] = before + heads;
      d_n_out[1] = 0;
    }
    if (dbg) {
      for (int k = 0; k < 5; ++k) dbg[g * 8 + k] = stamp[k];
      dbg[g * 8 + 5] = m;
      dbg[g * 8 + 6] = heads;
      dbg[g * 8 + 7] = narrow ? -m_pad : m_pad;
    }
  }
#undef VX_STAMP
}

}  // namespace

int voxel_direct_groups(int n) { return (n + VX_QUANTUM - 1) / VX_QUANTUM; }
size_t voxel_direct_scratch_ints(int n) { return (size_t)VX_BINS + 8 + 4 * (size_t)voxel_direct_groups(n); }

// bins: voxel_direct_scratch_ints(n) ints, ALL zero before the first call (the histogram is left zero); published:
// voxel_direct_groups(n) 64-bit words that NOTHING else ever writes, zero before the first call (a word counts as this
// call's when its upper half is this call's number: memory that once held anything else could pass for one); keys: n ints;
// relpos: n ints; comp: n 64-bit words; d_n_out: 2 ints (sum = cells written); status: 1 int, non-zero = not done, use the
// sort path.
// d_plan (with d_bbox6): the parameters are derived on the device from the encoded bounding box in d_bbox6 (launch_bbox's result,
// queued in front) -- minb / divb are not looked at, the host has not seen the box yet; 14 ints of device memory: the plan and,
// behind it, the box (d_bbox6 itself is left initialised for the next pass).
hipError_t launch_voxel_grid_direct(const float4* pts, int n, float inv_leaf, const int minb[3], const int divb[3], int* bins,
                                    unsigned long long* published, int* keys, int* relpos, unsigned long long* comp, float4* out,
                                    int* d_n_out, int* status, hipStream_t stream, unsigned long long* clear_word, int* d_bbox6,
                                    int* d_plan) {
  static_assert(sizeof(VoxelPlan) == 8 * sizeof(int), "the plan is read back as 8 ints");
  const VoxelPlan* plan = reinterpret_cast<const VoxelPlan*>(d_plan);
  static const int no3[3] = {0, 0, 0}, one3[3] = {1, 1, 1};
  if (plan) {
    hipLaunchKernelGGL(voxel_plan_kernel, dim3(1), dim3(64), 0, stream, d_bbox6, inv_leaf, reinterpret_cast<VoxelPlan*>(d_plan), d_n_out, status,
                       d_plan + 8);
    minb = no3;
    divb = one3;
  }
  const long long ncells = (long long)divb[0] * divb[1] * divb[2];
  // buckets of cpb consecutive cells, as many of the 8192 as the index space fills (a power-of-two bucket would leave up to
  // half of them unused -- 4350 for a raw scan at 0.2 m -- and the near-field buckets twice as full)
  const unsigned int cpb = (unsigned int)((ncells + VX_BINS - 1) / VX_BINS);
  const int nbins = plan ? VX_BINS : (int)((ncells - 1) / cpb) + 1;  // (with a plan: the kernels read theirs)
  const int groups = voxel_direct_groups(n);
  int* hist = bins;
  int4* group_range = reinterpret_cast<int4*>(bins + VX_BINS + 8);
  static std::atomic<unsigned int> call_counter{0};
  unsigned int epoch = ++call_counter;
  if (epoch == 0u) epoch = ++call_counter;  // (zero is what a fresh buffer holds)
  const int blocks = (n + VX_BLOCK * VX_PPT - 1) / (VX_BLOCK * VX_PPT);
  hipLaunchKernelGGL(voxel_hist_kernel, dim3(blocks), dim3(VX_BLOCK), 0, stream, pts, n, inv_leaf, minb[0], minb[1], minb[2],
                     divb[0], divb[0] * divb[1], cpb, nbins, keys, relpos, hist, plan);
  hipLaunchKernelGGL(voxel_scatter_kernel, dim3(blocks), dim3(VX_BLOCK), 0, stream, keys, relpos, n, cpb, nbins, hist, groups,
                     group_range, status, comp, plan);
  // ICPGPU_VOXEL_DEBUG=1 (development): phase time stamps of every group, the slowest ones printed
  static const bool debug = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_VOXEL_DEBUG"); return e && std::atoi(e) != 0; }();
  long long* dbg = nullptr;
  if (debug && hipMalloc(reinterpret_cast<void**>(&dbg), (size_t)groups * 8 * sizeof(long long)) == hipSuccess)
    (void)hipMemsetAsync(dbg, 0, (size_t)groups * 8 * sizeof(long long), stream);
  // ICPGPU_VOXEL_TEST_STALL=1 (tests): the first group publishes 30 ms late -- the others give up, the sort path takes over
  static const int test_stall = [] {
    const char* e = ICPGPU_DEV_ENV("ICPGPU_VOXEL_TEST_STALL");
    const int v = e ? std::atoi(e) : 0;
    if (v) std::fprintf(stderr, "[icpgpu] WARNING: ICPGPU_VOXEL_TEST_STALL is set -- every voxel-filter call stalls 30 ms and falls back to the "
                                "sort path (results unchanged; a test's switch, never a production setting)\n");
    return v;
  }();
  hipLaunchKernelGGL(voxel_group_kernel, dim3(groups), dim3(VX_BLOCK), 0, stream, pts, n, comp, group_range, published, epoch, out,
                     d_n_out, hist, nbins, status, dbg, test_stall, clear_word, plan);
  if (dbg) {
    std::vector<long long> h((size_t)groups * 8);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h.data(), dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipFree(dbg);
    long long first = 0, last = 0;
    std::vector<int> order;
    for (int g = 0; g < groups; ++g)
      if (h[(size_t)g * 8 + 4]) {
        order.push_back(g);
        if (!first || h[(size_t)g * 8] < first) first = h[(size_t)g * 8];
        if (h[(size_t)g * 8 + 4] > last) last = h[(size_t)g * 8 + 4];
      }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return h[(size_t)a * 8 + 4] - h[(size_t)a * 8] > h[(size_t)b * 8 + 4] - h[(size_t)b * 8]; });
    std::fprintf(stderr, "[voxel] %d live groups, first start -> last end %.2f us (100 MHz clock)\n", (int)order.size(), (last - first) * 0.01);
    for (size_t k = 0; k < order.size() && k < 4; ++k) {
      const long long* r = &h[(size_t)order[k] * 8];
      std::fprintf(stderr, "[voxel]   group %d: m %lld pad %lld (< 0: 32-bit words) heads %lld | starts at %.2f | load %.2f sort %.2f gather+scan+offset %.2f sums %.2f us\n",
                   order[k], r[5], r[7], r[6], (r[0] - first) * 0.01, (r[1] - r[0]) * 0.01, (r[2] - r[1]) * 0.01, (r[3] - r[2]) * 0.01, (r[4] - r[3]) * 0.01);
    }
  }
  return hipGetLastError();
}

size_t voxel_temp_bytes(int n) {
  const size_t a = radix_sort_scratch_ints(n), b = exclusive_scan_scratch_ints(n);
  return (a > b ? a : b) * sizeof(int) + 256;
}

// keys/vals: 2*n ints each (ping-pong), flags/slots: n ints each, d_n_out: 1 int (cells written).
hipError_t launch_voxel_grid(const float4* pts, int n, float inv_leaf, const int minb[3], const int divb[3], int* keys,
                             int* vals, int* flags, int* slots, void* temp, size_t temp_bytes, float4* out, int* d_n_out,
                             hipStream_t stream) {
  if (n <= 0) return hipMemsetAsync(d_n_out, 0, sizeof(int), stream);
  const int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks), dim3(256), 0, stream, pts, n, inv_leaf, minb[0], minb[1], minb[2], divb[0],
                     divb[0] * divb[1], keys, vals);
  // only the bits the cell indices of THIS cloud can have (+1: the sentinel of non-finite points has bit 30 set and must
  // still sort last): a raw scan at 0.2 m needs 25 of the 31, one digit pass fewer
  unsigned int end_bit = 1;
  const long long ncells = (long long)divb[0] * divb[1] * divb[2];
  while (end_bit < 31 && (1ll << end_bit) < ncells) ++end_bit;
  end_bit = end_bit < 31 ? end_bit + 1 : 31;
  // PCL tests the float extents for overflow and indexes with the integer ones: when those are a cell wider the topmost
  // cells' index wraps negative, and PCL's sort (signed) puts them first -- the sign bit must take part then
  if (ncells > 0x7FFFFFFFll) end_bit = 32;
  hipError_t e = launch_radix_sort_pairs(keys, vals, n, end_bit, static_cast<int*>(temp), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_flag_kernel, dim3(blocks), dim3(256), 0, stream, keys + n, n, flags);
  e = launch_exclusive_scan(flags, slots, n, static_cast<int*>(temp), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_centroid_kernel, dim3((n + VC_BLOCK - 1) / VC_BLOCK), dim3(VC_BLOCK), 0, stream, pts, keys + n,
                     vals + n, flags, slots, n, out);
  // number of cells = slots[n-1] + flags[n-1]
  e = hipMemcpyAsync(d_n_out, slots + (n - 1), sizeof(int), hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(d_n_out + 1, flags + (n - 1), sizeof(int), hipMemcpyDeviceToDevice, stream);
}

}  // namespace icpgpu
