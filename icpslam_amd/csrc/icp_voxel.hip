// icp_voxel.hip -- voxel-grid down-sampling on the device: SURVEY.md section 8(f2), the step right before the ICP path.
//
// Replaces pcl::VoxelGrid<PointXYZ>::filter as the reference calls it in IcpOdometer::voxelFilterCloud
// (/root/reference/src/icpslam/icp_odometer.cpp:96-101, leaf 0.2 m in /root/reference/config/icpslam.yaml:14):
// cell index = (floor(x/L) - min_b) . (1, div_x, div_x*div_y); one output point per occupied cell = arithmetic mean of
// its points; output ordered by ascending cell index.  HBM-bound: 16 B read + 8 B (key, index) written per point, a
// 4-pass 8-bit LSD radix sort over the 31-bit keys (rocPRIM device primitive -- stable, so points of a cell stay in
// input order and the float32 mean is accumulated in exactly the oracle's order), boundary flags + scan, one gather
// pass.  Same float32 arithmetic as oracle/icp_oracle.c::orc_voxel_grid, so the output is bit-identical to it.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

}  // namespace

size_t voxel_temp_bytes(int n) {
  size_t a = 0, b = 0;
  int* ip = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, a, ip, ip, ip, ip, (size_t)n, 0, 31, (hipStream_t) nullptr);
  (void)rocprim::exclusive_scan(nullptr, b, ip, ip, 0, (size_t)n, rocprim::plus<int>(), (hipStream_t) nullptr);
  return (a > b ? a : b) + 256;
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
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys + n, vals, vals + n, (size_t)n, 0, end_bit, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_flag_kernel, dim3(blocks), dim3(256), 0, stream, keys + n, n, flags);
  e = rocprim::exclusive_scan(temp, temp_bytes, flags, slots, 0, (size_t)n, rocprim::plus<int>(), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxel_centroid_kernel, dim3((n + VC_BLOCK - 1) / VC_BLOCK), dim3(VC_BLOCK), 0, stream, pts, keys + n,
                     vals + n, flags, slots, n, out);
  // number of cells = slots[n-1] + flags[n-1]
  e = hipMemcpyAsync(d_n_out, slots + (n - 1), sizeof(int), hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(d_n_out + 1, flags + (n - 1), sizeof(int), hipMemcpyDeviceToDevice, stream);
}

}  // namespace icpgpu
