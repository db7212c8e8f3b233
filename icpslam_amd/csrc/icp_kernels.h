// icp_kernels.h -- launch wrappers for the gfx950 kernels of the ICP hot path (internal).
//
// Rows of SURVEY.md §8(a) implemented here:
//   a2  nearest-neighbour correspondence search   -> launch_nn_brute()
//   a3  outlier rejection (d2 <= r^2 predicate)   -> fused into launch_reduce()
//   a4  centroid / cross-covariance reduction     -> launch_reduce()
//   a6  point-set transform                       -> fused into a2/a4 loads; launch_transform() for the output cloud
//   a9  fitness score                             -> a2 + launch_reduce() with an open threshold
// Reference call sites these replace: /root/reference/src/icpslam/icp_odometer.cpp:198-201,
// /root/reference/src/icpslam/octree_mapper.cpp:114-117 (the arithmetic itself lives in PCL).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpgpu {

// Rigid transform as 12 floats (row-major 3x4) so that it travels in SGPRs as a kernel argument.
struct Xform {
  float m[12];
};

// 64-bit correspondence key: (float bits of d2) << 32 | target index. For d2 >= 0 the unsigned order of
// the key is (d2, index) lexicographic, so a u64 min merges partial searches and breaks ties on the lowest index.
static constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;

static constexpr int kReduceTerms = 17;   // n, Sp(3), Sq(3), Sqp(9), Sd2
static constexpr int kMaxReduceBlocks = 1024;

struct NnPlan {
  int variant;        // 0 = LDS-tiled, 1 = scalar-load (SGPR broadcast)
  int splits;         // target splits (grid.y)
  int tgt_per_split;  // multiple of the LDS tile
  int grid_x;
};

NnPlan plan_nn_brute(int n_s, int n_t, int variant, int num_cus);

// keys[i] = min over target of (d2, j) for p = T*src[i]. When plan.splits > 1 keys must be pre-filled with kEmptyKey.
hipError_t launch_nn_brute(const float4* src, int n_s, const float4* tgt, int n_t, const Xform& T, const NnPlan& plan,
                           unsigned long long* keys, hipStream_t stream);

// partials: [blocks][17] doubles, sums_out: 17 doubles (device). Deterministic (fixed order) two-stage reduction.
hipError_t launch_reduce(const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, const Xform& T,
                         float d2_threshold, double* partials, double* sums_out, hipStream_t stream);

hipError_t launch_transform(const float4* src, int n_s, const Xform& T, float4* out, hipStream_t stream);

hipError_t launch_fill_keys(unsigned long long* keys, int n, hipStream_t stream);

// keys -> (idx, d2) arrays for the kernel-level C-ABI entry point.
hipError_t launch_unpack_keys(const unsigned long long* keys, int n, int32_t* idx, float* d2, hipStream_t stream);

}  // namespace icpgpu
