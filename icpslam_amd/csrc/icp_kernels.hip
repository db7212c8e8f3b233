// icp_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the ICP correspondence / reduction hot path.
//
// What they replace (arithmetic lives in PCL, reached from the reference at
// /root/reference/src/icpslam/icp_odometer.cpp:198-201 and src/icpslam/octree_mapper.cpp:114-117):
//   a2  CorrespondenceEstimation::determineCorrespondences  -> nn_brute_kernel
//   a3  "if (distance > max_dist_sqr) continue"              -> predicate inside reduce_kernel
//   a4  TransformationEstimationSVD / Eigen::umeyama sums     -> reduce_kernel + reduce_final_kernel
//   a6  transformPointCloud                                   -> fused into every load of the source; transform_kernel for the output
//
// Arithmetic contract (identical, operation for operation, in oracle/icp_oracle.c; this file is compiled with
// -ffp-contract=off so the only fused operations are the explicit __builtin_fmaf calls):
//   p.x = fma(m2, z, fma(m1, y, fma(m0, x, m3)))          d2 = fma(dz, dz, fma(dy, dy, dx*dx)),  dx = q.x - p.x
//
// Data layout in HBM: clouds stay in the caller's pcl::PointXYZ layout (float4 AoS, 16 B/point) so a wave reads
// 1 KiB per global_load_dwordx4; correspondences are 8-byte packed keys (d2 bits << 32 | index).
#include "icp_kernels.h"

#include <math.h>

namespace icpgpu {
namespace {

constexpr int NN_BLOCK = 256;  // 4 waves
constexpr int NN_R = 4;        // source points held in registers per lane
constexpr int NN_TILE = 1024;  // target points per LDS tile (16 KiB)
constexpr int NN_CHUNK = 8;    // targets folded with v_min3 before one index-tracking compare

__device__ __forceinline__ void xform_point(const Xform& T, float x, float y, float z, float& px, float& py, float& pz) {
  px = __builtin_fmaf(T.m[2], z, __builtin_fmaf(T.m[1], y, __builtin_fmaf(T.m[0], x, T.m[3])));
  py = __builtin_fmaf(T.m[6], z, __builtin_fmaf(T.m[5], y, __builtin_fmaf(T.m[4], x, T.m[7])));
  pz = __builtin_fmaf(T.m[10], z, __builtin_fmaf(T.m[9], y, __builtin_fmaf(T.m[8], x, T.m[11])));
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// ------------------------------------------------------------------------------------------------------------
// a2: brute-force nearest neighbour.
//   grid.x : blocks of NN_BLOCK*NN_R source points (each lane keeps NN_R transformed points in VGPRs)
//   grid.y : target splits; partial results are merged with one 64-bit atomic min per (source point, split)
//   VARIANT 0: target tiles staged in LDS, read back as wave-uniform ds_read_b128 broadcasts
//   VARIANT 1: target streamed through the scalar cache (s_load), operands arrive as SGPRs
// Index tracking is done per chunk of NN_CHUNK targets: min3-fold the chunk's distances, one compare + two selects
// per chunk, then re-scan the winning chunk once at the end for the exact (lowest) index.
// ------------------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(NN_BLOCK) void nn_brute_kernel(const float4* __restrict__ src, int n_s,
                                                            const float4* __restrict__ tgt, int n_t, Xform T,
                                                            int tgt_per_split, int splits,
                                                            unsigned long long* __restrict__ keys) {
  const int tid = threadIdx.x;
  const int base = blockIdx.x * (NN_BLOCK * NN_R);

  float px[NN_R], py[NN_R], pz[NN_R], best[NN_R];
  int bchunk[NN_R];
#pragma unroll
  for (int r = 0; r < NN_R; ++r) {
    const int i = base + r * NN_BLOCK + tid;
    const float4 s = src[i < n_s ? i : n_s - 1];
    xform_point(T, s.x, s.y, s.z, px[r], py[r], pz[r]);
    best[r] = INFINITY;
    bchunk[r] = -1;
  }

  const int j0 = blockIdx.y * tgt_per_split;
  const int j1 = min(n_t, j0 + tgt_per_split);

  if constexpr (VARIANT == 0) {
    __shared__ float4 tile[NN_TILE];
    for (int jt = j0; jt < j1; jt += NN_TILE) {
      __syncthreads();
#pragma unroll
      for (int k = tid; k < NN_TILE; k += NN_BLOCK) {
        const int j = jt + k;
        tile[k] = (j < j1) ? tgt[j] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
      }
      __syncthreads();
      const int lim = min(NN_TILE, j1 - jt);
#pragma unroll 2
      for (int c = 0; c < lim; c += NN_CHUNK) {
        float4 q[NN_CHUNK];
#pragma unroll
        for (int u = 0; u < NN_CHUNK; ++u) q[u] = tile[c + u];
#pragma unroll
        for (int r = 0; r < NN_R; ++r) {
          float m = dist2(q[0].x, q[0].y, q[0].z, px[r], py[r], pz[r]);
#pragma unroll
          for (int u = 1; u < NN_CHUNK; ++u) m = fminf(m, dist2(q[u].x, q[u].y, q[u].z, px[r], py[r], pz[r]));
          if (m < best[r]) {
            best[r] = m;
            bchunk[r] = jt + c;
          }
        }
      }
    }
  } else {
    for (int jc = j0; jc < j1; jc += NN_CHUNK) {
      float4 q[NN_CHUNK];
#pragma unroll
      for (int u = 0; u < NN_CHUNK; ++u) {
        const int j = jc + u;
        q[u] = tgt[j < n_t ? j : n_t - 1];
        if (j >= j1) q[u].x = INFINITY;
      }
#pragma unroll
      for (int r = 0; r < NN_R; ++r) {
        float m = dist2(q[0].x, q[0].y, q[0].z, px[r], py[r], pz[r]);
#pragma unroll
        for (int u = 1; u < NN_CHUNK; ++u) m = fminf(m, dist2(q[u].x, q[u].y, q[u].z, px[r], py[r], pz[r]));
        if (m < best[r]) {
          best[r] = m;
          bchunk[r] = jc;
        }
      }
    }
  }

#pragma unroll
  for (int r = 0; r < NN_R; ++r) {
    const int i = base + r * NN_BLOCK + tid;
    if (i >= n_s) continue;
    unsigned long long key = kEmptyKey;
    if (bchunk[r] >= 0) {
      int idx = -1;
#pragma unroll
      for (int u = NN_CHUNK - 1; u >= 0; --u) {  // descending so the lowest matching index is kept
        const int j = bchunk[r] + u;
        if (j < j1) {
          const float4 q = tgt[j];
          if (dist2(q.x, q.y, q.z, px[r], py[r], pz[r]) == best[r]) idx = j;
        }
      }
      key = ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned int)idx;
    }
    if (splits > 1)
      atomicMin(&keys[i], key);
    else
      keys[i] = key;
  }
}

__global__ void fill_keys_kernel(unsigned long long* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = kEmptyKey;
}

__global__ void unpack_keys_kernel(const unsigned long long* __restrict__ keys, int n, int32_t* __restrict__ idx,
                                   float* __restrict__ d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const bool empty = (unsigned int)k == 0xFFFFFFFFu;
  idx[i] = empty ? -1 : (int32_t)(unsigned int)k;
  d2[i] = empty ? INFINITY : __uint_as_float((unsigned int)(k >> 32));
}

// ------------------------------------------------------------------------------------------------------------
// a3 + a4: rejection predicate + 17-term reduction in double, deterministic order.
//   stage 1: grid-stride over source points, per-lane double accumulators, wave shuffle tree, LDS across the
//            4 waves, one 17-double partial per block
//   stage 2: one workgroup of 17 waves (one per term) sums the partials in a fixed order
// HBM traffic: 8 B key + 16 B source + 16 B gathered target per accepted point.
// ------------------------------------------------------------------------------------------------------------
constexpr int RED_BLOCK = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(RED_BLOCK) void reduce_kernel(const float4* __restrict__ src, int n_s,
                                                           const float4* __restrict__ tgt,
                                                           const unsigned long long* __restrict__ keys, Xform T,
                                                           float thr, double* __restrict__ partials) {
  double acc[kReduceTerms];
#pragma unroll
  for (int k = 0; k < kReduceTerms; ++k) acc[k] = 0.0;

  for (int i = blockIdx.x * RED_BLOCK + threadIdx.x; i < n_s; i += gridDim.x * RED_BLOCK) {
    const unsigned long long key = keys[i];
    const unsigned int j = (unsigned int)key;
    const float d2 = __uint_as_float((unsigned int)(key >> 32));
    if (j != 0xFFFFFFFFu && d2 <= thr) {
      const float4 s = src[i];
      float px, py, pz;
      xform_point(T, s.x, s.y, s.z, px, py, pz);
      const float4 q = tgt[j];
      const double p[3] = {(double)px, (double)py, (double)pz};
      const double qq[3] = {(double)q.x, (double)q.y, (double)q.z};
      acc[0] += 1.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        acc[1 + a] += p[a];
        acc[4 + a] += qq[a];
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[7 + 3 * a + b] += qq[a] * p[b];
      }
      acc[16] += (double)d2;
    }
  }

  __shared__ double wsum[RED_BLOCK / 64][kReduceTerms];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kReduceTerms; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) wsum[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kReduceTerms) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < RED_BLOCK / 64; ++w) v += wsum[w][threadIdx.x];
    partials[(size_t)blockIdx.x * kReduceTerms + threadIdx.x] = v;
  }
}

// stage 2: 16 waves; wave w owns terms w, w+16; lane l adds partials l, l+64, ... in that order, then a fixed
// shuffle tree -> bitwise reproducible run to run.
__global__ __launch_bounds__(1024) void reduce_final_kernel(const double* __restrict__ partials, int n_blocks,
                                                            double* __restrict__ sums) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = wave; k < kReduceTerms; k += 16) {
    double v = 0.0;
    for (int b = lane; b < n_blocks; b += 64) v += partials[(size_t)b * kReduceTerms + k];
    v = wave_sum(v);
    if (lane == 0) sums[k] = v;
  }
}

// a6: output cloud. 32 B/point of HBM traffic, float4 in / float4 out.
__global__ __launch_bounds__(256) void transform_kernel(const float4* __restrict__ src, int n, Xform T,
                                                        float4* __restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 s = src[i];
    float4 o;
    xform_point(T, s.x, s.y, s.z, o.x, o.y, o.z);
    o.w = 1.0f;
    out[i] = o;
  }
}

}  // namespace

NnPlan plan_nn_brute(int n_s, int n_t, int variant, int num_cus) {
  NnPlan p;
  p.variant = variant;
  p.grid_x = (n_s + NN_BLOCK * NN_R - 1) / (NN_BLOCK * NN_R);
  if (p.grid_x < 1) p.grid_x = 1;
  // aim for ~8 resident 256-thread workgroups per CU so the tail imbalance stays small; never split below one tile
  const int want_wgs = num_cus * 8;
  int splits = (want_wgs + p.grid_x - 1) / p.grid_x;
  const int max_splits = (n_t + NN_TILE - 1) / NN_TILE;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int per = (n_t + splits - 1) / splits;
  per = ((per + NN_TILE - 1) / NN_TILE) * NN_TILE;
  if (per < NN_TILE) per = NN_TILE;
  p.tgt_per_split = per;
  p.splits = (n_t + per - 1) / per;
  if (p.splits < 1) p.splits = 1;
  return p;
}

hipError_t launch_nn_brute(const float4* src, int n_s, const float4* tgt, int n_t, const Xform& T, const NnPlan& plan,
                           unsigned long long* keys, hipStream_t stream) {
  if (n_s <= 0) return hipSuccess;
  if (n_t <= 0) return launch_fill_keys(keys, n_s, stream);
  dim3 grid(plan.grid_x, plan.splits), block(NN_BLOCK);
  if (plan.variant == 1)
    hipLaunchKernelGGL(nn_brute_kernel<1>, grid, block, 0, stream, src, n_s, tgt, n_t, T, plan.tgt_per_split,
                       plan.splits, keys);
  else
    hipLaunchKernelGGL(nn_brute_kernel<0>, grid, block, 0, stream, src, n_s, tgt, n_t, T, plan.tgt_per_split,
                       plan.splits, keys);
  return hipGetLastError();
}

hipError_t launch_reduce(const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, const Xform& T,
                         float thr, double* partials, double* sums_out, hipStream_t stream) {
  int blocks = (n_s + RED_BLOCK - 1) / RED_BLOCK;
  if (blocks > kMaxReduceBlocks) blocks = kMaxReduceBlocks;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(RED_BLOCK), 0, stream, src, n_s, tgt, keys, T, thr, partials);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(1024), 0, stream, partials, blocks, sums_out);
  return hipGetLastError();
}

hipError_t launch_transform(const float4* src, int n_s, const Xform& T, float4* out, hipStream_t stream) {
  if (n_s <= 0) return hipSuccess;
  int blocks = (n_s + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(256), 0, stream, src, n_s, T, out);
  return hipGetLastError();
}

hipError_t launch_fill_keys(unsigned long long* keys, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n);
  return hipGetLastError();
}

hipError_t launch_unpack_keys(const unsigned long long* keys, int n, int32_t* idx, float* d2, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n, idx, d2);
  return hipGetLastError();
}

}  // namespace icpgpu
