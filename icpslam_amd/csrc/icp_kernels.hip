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

#include "icp_device.h"
#include "icp_grid_device.h"

namespace icpgpu {
namespace {

constexpr int NN_BLOCK = 256;  // 4 waves
constexpr int NN_TILE = 1024;  // target points per LDS tile (16 KiB)
constexpr int NN_CHUNK = 8;    // targets folded with v_min3 before one index-tracking compare

// ------------------------------------------------------------------------------------------------------------
// a2: brute-force nearest neighbour.
//   grid.x : blocks of NN_BLOCK*R source points (each lane keeps R transformed points in VGPRs)
//   grid.y : target splits; partial results are merged with one 64-bit atomic min per (source point, split)
//   VARIANT 0: target tiles staged in LDS, read back as wave-uniform ds_read broadcasts, scalar f32 VALU
//   VARIANT 1: target streamed through the scalar cache (s_load), operands arrive as SGPRs, scalar f32 VALU
//   VARIANT 2: as 0, but two source points per lane share each v_pk_add/v_pk_mul/v_pk_fma_f32 (same IEEE results)
// Index tracking is done per chunk of NN_CHUNK targets: min3-fold the chunk's distances, one compare + two selects
// per chunk, then re-scan the winning chunk once at the end for the exact (lowest) index.
// ------------------------------------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2v dist2_pk(float qx, float qy, float qz, float2v px, float2v py, float2v pz) {
  const float2v dx = qx - px, dy = qy - py, dz = qz - pz;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

// fold one chunk of NN_CHUNK targets (q) into the running (best, bchunk) of R source points
template <int R, bool PACKED>
__device__ __forceinline__ void fold_chunk(const float4 (&q)[NN_CHUNK], const float (&px)[R], const float (&py)[R],
                                           const float (&pz)[R], float (&best)[R], int (&bchunk)[R], int chunk_start) {
  if constexpr (PACKED) {
    static_assert(R % 2 == 0, "packed variant needs an even number of points per lane");
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      const float2v x2 = {px[r], px[r + 1]}, y2 = {py[r], py[r + 1]}, z2 = {pz[r], pz[r + 1]};
      float2v m = dist2_pk(q[0].x, q[0].y, q[0].z, x2, y2, z2);
#pragma unroll
      for (int u = 1; u < NN_CHUNK; ++u) {
        const float2v d = dist2_pk(q[u].x, q[u].y, q[u].z, x2, y2, z2);
        m.x = fminf(m.x, d.x);
        m.y = fminf(m.y, d.y);
      }
      if (m.x < best[r]) {
        best[r] = m.x;
        bchunk[r] = chunk_start;
      }
      if (m.y < best[r + 1]) {
        best[r + 1] = m.y;
        bchunk[r + 1] = chunk_start;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float m = dist2(q[0].x, q[0].y, q[0].z, px[r], py[r], pz[r]);
#pragma unroll
      for (int u = 1; u < NN_CHUNK; ++u) m = fminf(m, dist2(q[u].x, q[u].y, q[u].z, px[r], py[r], pz[r]));
      if (m < best[r]) {
        best[r] = m;
        bchunk[r] = chunk_start;
      }
    }
  }
}

template <int VARIANT, int R>
__global__ __launch_bounds__(NN_BLOCK) void nn_brute_kernel(const float4* __restrict__ src, int n_s,
                                                            const float4* __restrict__ tgt, int n_t, Xform T,
                                                            int tgt_per_split, int splits,
                                                            unsigned long long* __restrict__ keys,
                                                            const int* __restrict__ list) {
  // `list` (optional): the n_s logical source points are src[list[k]] and results go to keys[list[k]]
  const int tid = threadIdx.x;
  const int base = blockIdx.x * (NN_BLOCK * R);

  float px[R], py[R], pz[R], best[R];
  int bchunk[R], srci[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = min(base + r * NN_BLOCK + tid, n_s - 1);
    srci[r] = list ? list[k] : k;
    const float4 s = src[srci[r]];
    xform_point(T, s.x, s.y, s.z, px[r], py[r], pz[r]);
    best[r] = INFINITY;
    bchunk[r] = -1;
  }

  const int j0 = blockIdx.y * tgt_per_split;
  const int j1 = min(n_t, j0 + tgt_per_split);

  if constexpr (VARIANT != 1) {
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
        fold_chunk<R, VARIANT == 2>(q, px, py, pz, best, bchunk, jt + c);
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
      fold_chunk<R, false>(q, px, py, pz, best, bchunk, jc);
    }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (base + r * NN_BLOCK + tid >= n_s) continue;
    const int i = srci[r];
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

// Completion for a HANDFUL of listed points (the few far outliers a gate-less search leaves behind): lane = target point,
// the listed queries are a wave-uniform outer loop.  The tiled kernel above costs a full pass over the target whatever
// the number of queries (62 us at 50k targets for 9 points); this one is launch-bound.
constexpr int kFewList = 64;
constexpr int kFewPerLane = 4;  // target points a lane keeps in registers while the listed queries go by
// The lane's target points are read ONCE and stay in registers; the listed queries are the inner loop (their
// coordinates are wave-uniform: scalar loads).  Per query a wave minimum (distance bits by DPP, index only among ties),
// one LDS slot per wave and query, and at the very end one atomicMin per workgroup and query.  59 queries over 200k
// targets: 8 us (45 us when every query was a separate pass with two barriers).  The number of listed queries may come
// from device memory (count_ptr): getFitnessScore then needs no host round trip between the grid search and its
// completion; more than kFewList queries leave the keys alone and report the count, the host takes the tiled kernel.
__global__ __launch_bounds__(256) void nn_brute_few_kernel(const float4* __restrict__ src, const int* __restrict__ list,
                                                           const int* __restrict__ count_ptr, int n_list_host,
                                                           const float4* __restrict__ tgt, int n_t, Xform T,
                                                           unsigned long long* __restrict__ keys, int* __restrict__ count_out) {
  const int n_list = count_ptr ? *count_ptr : n_list_host;
  if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = n_list;
  if (n_list <= 0 || n_list > kFewList) return;
  __shared__ unsigned long long wbest[kFewList][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)threadIdx.x < n_list) {
#pragma unroll
    for (int w = 0; w < 4; ++w) wbest[threadIdx.x][w] = kEmptyKey;
  }
  __syncthreads();
  const int span = gridDim.x * 256 * kFewPerLane;
  for (int base = 0; base < n_t; base += span) {
    float4 q[kFewPerLane];
    int j[kFewPerLane];
#pragma unroll
    for (int u = 0; u < kFewPerLane; ++u) {
      j[u] = base + (u * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
      q[u] = tgt[min(j[u], n_t - 1)];
    }
    for (int qi = 0; qi < n_list; ++qi) {
      const float4 s = src[list[qi]];
      float px, py, pz;
      xform_point(T, s.x, s.y, s.z, px, py, pz);
      unsigned long long best = kEmptyKey;
#pragma unroll
      for (int u = 0; u < kFewPerLane; ++u) {
        const float d = dist2(q[u].x, q[u].y, q[u].z, px, py, pz);
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)j[u];
        if (j[u] < n_t && d < INFINITY && key < best) best = key;  // like the tiled kernel: NaN and inf distances never win
      }
      const unsigned int dbits = (unsigned int)(best >> 32);
      const unsigned int dmin = wave_min_u32(dbits);
      if (dmin == 0xFFFFFFFFu) continue;  // wave-uniform: no lane has a candidate
      const unsigned int imin = wave_min_u32(dbits == dmin ? (unsigned int)best : 0xFFFFFFFFu);
      if (lane == 0) {
        const unsigned long long k = ((unsigned long long)dmin << 32) | imin;
        if (k < wbest[qi][wave]) wbest[qi][wave] = k;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_list) {
    unsigned long long b = wbest[threadIdx.x][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = wbest[threadIdx.x][w] < b ? wbest[threadIdx.x][w] : b;
    if (b != kEmptyKey) atomicMin(&keys[list[threadIdx.x]], b);
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
//   stage 2: 17 workgroups (one per term) sum the partials in a fixed order
// HBM traffic: 8 B key + 16 B source + 16 B gathered target per accepted point.
// ------------------------------------------------------------------------------------------------------------
constexpr int RED_BLOCK = 256;

__global__ __launch_bounds__(RED_BLOCK) void reduce_kernel(const float4* __restrict__ src, int n_s,
                                                           const float4* __restrict__ tgt,
                                                           const unsigned long long* __restrict__ keys, Xform T,
                                                           float thr, double* __restrict__ partials) {
  double acc[kReduceTerms];
#pragma unroll
  for (int k = 0; k < kReduceTerms; ++k) acc[k] = 0.0;

  // Four points per trip, their loads issued together (keys, then sources and the targets the keys name), accumulated in
  // the order of one at a time: key -> target is a dependent pair of reads, and one point after the other made a lane's four
  // points eight serial memory round trips.
  const int stride = gridDim.x * RED_BLOCK;
  for (int i0 = blockIdx.x * RED_BLOCK + threadIdx.x; i0 < n_s; i0 += 4 * stride) {
    unsigned long long key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) key[u] = i0 + u * stride < n_s ? keys[i0 + u * stride] : kEmptyKey;
    float4 s[4], q[4];
    bool use[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned int j = (unsigned int)key[u];
      use[u] = j != 0xFFFFFFFFu && __uint_as_float((unsigned int)(key[u] >> 32)) <= thr;
      s[u] = src[min(i0 + u * stride, n_s - 1)];
      q[u] = tgt[use[u] ? j : 0u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!use[u]) continue;
      float px, py, pz;
      xform_point(T, s[u].x, s[u].y, s[u].z, px, py, pz);
      accumulate_pair(acc, px, py, pz, q[u].x, q[u].y, q[u].z, __uint_as_float((unsigned int)(key[u] >> 32)));
    }
  }
  block_reduce_store<RED_BLOCK / 64>(acc, partials);
}

// stage 2: one 256-thread workgroup per term; thread t adds partials t, t+256, ... in that order (independent loads),
// then a fixed shuffle + LDS tree -> bitwise reproducible run to run.
// `flags` (optional) lets a host that polls pinned memory see the result without a stream synchronisation: term k's
// workgroup publishes the pair {sum, seq} at flags[2k], flags[2k + 1] (then `sums` is not written).
// term_major: partials[k * n_blocks + b] (coalesced reads here; the fused grid search writes this layout) instead of
// partials[b * 17 + k].
// 1024 threads per term: a 200k-point sweep's 3128 partials are ONE batch of loads per thread (one memory round trip).
constexpr int RF_BLOCK = 1024;
__device__ __forceinline__ void reduce_final_body(const double* __restrict__ partials, int n_blocks, int term_major,
                                                  double* __restrict__ sums, unsigned long long* flags, unsigned long long seq) {
  const int k = blockIdx.x;
  double v = 0.0;
  // Eight loads in flight, then the eight additions in the same order as one at a time (same bits): a load -> wait -> add
  // loop made this kernel 12 serial memory round trips long for a 200k-point sweep (5.1 us; the sums are on every ICP
  // iteration's critical path).
  const double* __restrict__ base = term_major ? partials + (size_t)k * n_blocks : partials + k;
  const size_t step = term_major ? 1 : (size_t)kReduceTerms;
  for (int b0 = threadIdx.x; b0 < n_blocks; b0 += 8 * RF_BLOCK) {
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + u * RF_BLOCK;
      x[u] = b < n_blocks ? base[(size_t)b * step] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (b0 + u * RF_BLOCK < n_blocks) v += x[u];  // (no "+ 0.0": -0.0 partials must add up as before)
  }
  v = wave_sum(v);
  __shared__ double w[RF_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double sum = 0.0;  // a fixed tree over the waves' sums
    {
      double t[RF_BLOCK / 64];
#pragma unroll
      for (int i = 0; i < RF_BLOCK / 64; ++i) t[i] = w[i];
#pragma unroll
      for (int span = 1; span < RF_BLOCK / 64; span <<= 1)
#pragma unroll
        for (int i = 0; i + span < RF_BLOCK / 64; i += 2 * span) t[i] += t[i + span];
      sum = t[0];
    }
    if (flags) {
      // the host mailbox: {sum, seq} as ONE 16-byte store written through to system memory -- a pair is never seen
      // half-written, so the host needs no flag behind a system-scope release (a write-back + a wait for the sum's
      // acknowledgement before the flag may leave: ~1 us of every ICP iteration)
      store_result_pair(flags + 2 * k, (unsigned long long)__double_as_longlong(sum), seq);
    } else {
      sums[k] = sum;
    }
  }
}

__global__ __launch_bounds__(RF_BLOCK) void reduce_final_kernel(const double* __restrict__ partials, int n_blocks, int term_major,
                                                           double* __restrict__ sums, unsigned long long* flags,
                                                           unsigned long long seq) {
  reduce_final_body(partials, n_blocks, term_major, sums, flags, seq);
}

// the final reductions of a lock-step sweep: pair step.slot[blockIdx.y], 17 workgroups each, into that pair's mailbox
__global__ __launch_bounds__(RF_BLOCK) void reduce_final_batch_kernel(const BatchPair* __restrict__ pairs, BatchStep step) {
  const BatchPair& d = pairs[step.slot[blockIdx.y]];
  reduce_final_body(d.partials, d.blocks, 1, nullptr, d.flags, step.seq[blockIdx.y]);
}

// a6: output cloud. 32 B/point of HBM traffic, float4 in / float4 out.
// Four independent 16-B loads in flight per lane (4 KiB per wave) before the first store; streaming stores.
__global__ __launch_bounds__(256) void transform_kernel(const float4* __restrict__ src, int n, Xform T,
                                                        float4* __restrict__ out) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int stride = gridDim.x * 256;
  int i = blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    float4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = src[i + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v4f o;
      float ox, oy, oz;
      xform_point(T, s[k].x, s[k].y, s[k].z, ox, oy, oz);
      o.x = ox; o.y = oy; o.z = oz; o.w = 1.0f;
      __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(out + i + k * stride));
    }
  }
  for (; i < n; i += stride) {
    const float4 s = src[i];
    float4 o;
    xform_point(T, s.x, s.y, s.z, o.x, o.y, o.z);
    o.w = 1.0f;
    out[i] = o;
  }
}

}  // namespace

static int points_per_lane(int variant) { return variant >= 10 ? 8 : 4; }

NnPlan plan_nn_brute(int n_s, int n_t, int variant, int num_cus) {
  NnPlan p;
  p.variant = variant;
  const int per_block = NN_BLOCK * points_per_lane(variant);
  p.grid_x = (n_s + per_block - 1) / per_block;
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
#define ICP_LAUNCH_NN(V, R)                                                                                      \
  hipLaunchKernelGGL((nn_brute_kernel<V, R>), grid, block, 0, stream, src, n_s, tgt, n_t, T, plan.tgt_per_split, \
                     plan.splits, keys, (const int*)nullptr)
  switch (plan.variant) {
    case 1: ICP_LAUNCH_NN(1, 4); break;
    case 2: ICP_LAUNCH_NN(2, 4); break;
    case 10: ICP_LAUNCH_NN(0, 8); break;
    case 12: ICP_LAUNCH_NN(2, 8); break;
    default: ICP_LAUNCH_NN(0, 4); break;
  }
#undef ICP_LAUNCH_NN
  return hipGetLastError();
}

hipError_t launch_nn_brute_few(const float4* src, const int* list, const int* count_ptr, int n_list, const float4* tgt, int n_t,
                               const Xform& T, unsigned long long* keys, int* count_out, hipStream_t stream) {
  if (n_t <= 0) return hipSuccess;
  // (keys of listed points are empty on entry: the atomic-min merge is valid)
  int blocks = (n_t + 256 * kFewPerLane - 1) / (256 * kFewPerLane);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(nn_brute_few_kernel, dim3(blocks), dim3(256), 0, stream, src, list, count_ptr, n_list, tgt, n_t, T, keys,
                     count_out);
  return hipGetLastError();
}

hipError_t launch_nn_brute_list(const float4* src, const int* list, int n_list, const float4* tgt, int n_t,
                                const Xform& T, int num_cus, unsigned long long* keys, hipStream_t stream) {
  if (n_list <= 0 || n_t <= 0) return hipSuccess;
  if (n_list <= kFewList) return launch_nn_brute_few(src, list, nullptr, n_list, tgt, n_t, T, keys, nullptr, stream);
  const NnPlan plan = plan_nn_brute(n_list, n_t, 0, num_cus);
  dim3 grid(plan.grid_x, plan.splits), block(NN_BLOCK);
  // keys of listed points are empty on entry, so the atomic-min merge is always valid (splits is forced > 1)
  hipLaunchKernelGGL((nn_brute_kernel<0, 4>), grid, block, 0, stream, src, n_list, tgt, n_t, T, plan.tgt_per_split, 2,
                     keys, list);
  return hipGetLastError();
}

hipError_t launch_reduce(const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, const Xform& T,
                         float thr, double* partials, double* sums_out, unsigned long long* flags, unsigned long long seq,
                         hipStream_t stream) {
  // ~4 points per lane: the 17-term workgroup reduction at the end costs more than a point does
  int blocks = (n_s + 4 * RED_BLOCK - 1) / (4 * RED_BLOCK);
  if (blocks > kMaxReduceBlocks) blocks = kMaxReduceBlocks;
  if (blocks < 1) blocks = 1;
  // reduce_kernel reads tgt[0] for every slot it does not use (its loads are issued before the keys are judged).  An EMPTY
  // target never allocated a buffer: every key is empty then, and the unused reads go to the source instead of address 0.
  if (!tgt) tgt = src;
  hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(RED_BLOCK), 0, stream, src, n_s, tgt, keys, T, thr, partials);
  return launch_reduce_final(partials, blocks, false, sums_out, flags, seq, stream);
}

// Start-up self-test of the result mailbox (icpgpu_create, once per device and process): one lane stores `rounds` pairs
// {bits_i, tag(i, bits_i)} into ONE slot of mapped host memory, a fraction of a microsecond apart, while the host reads the slot
// as fast as it can; a pair the host finds inconsistent although its tag did not move means the 16-byte store is not seen whole
// on this platform -- the context then uses the release form (icp_kernels.h).
__global__ void mailbox_selftest_kernel(unsigned long long* pair, int rounds) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 1; i <= rounds; ++i) {
    store_result_pair(pair, (unsigned long long)i * 0x9E3779B97F4A7C15ull, (unsigned long long)i);
    __builtin_amdgcn_s_sleep(16);
  }
}
// A handful of ints from device memory into the host mailbox as result pairs {value, tag(number)}: what a grid build or the voxel
// filter needs from the device between two of its kernels.  The host polls the pairs; against hipMemcpyAsync + hipStreamSynchronize
// that is 10 instead of 16 us per hand-over (scripts/probes/roundtrip_probe.cpp).
__global__ void post_ints_kernel(const int* __restrict__ src, int n, unsigned long long* pairs, unsigned long long number) {
  if ((int)threadIdx.x < n && blockIdx.x == 0) store_result_pair(pairs + 2 * threadIdx.x, (unsigned long long)(unsigned int)src[threadIdx.x], number);
}
hipError_t launch_post_ints(const int* d_src, int n, unsigned long long* pairs_dev, unsigned long long number, hipStream_t stream) {
  hipLaunchKernelGGL(post_ints_kernel, dim3(1), dim3(64), 0, stream, d_src, n, pairs_dev, number);
  return hipGetLastError();
}
hipError_t launch_mailbox_selftest(unsigned long long* pair_dev, int rounds, hipStream_t stream) {
  hipLaunchKernelGGL(mailbox_selftest_kernel, dim3(1), dim3(64), 0, stream, pair_dev, rounds);
  return hipGetLastError();
}

hipError_t launch_reduce_final(const double* partials, int n_blocks, bool term_major, double* sums_out,
                               unsigned long long* flags, unsigned long long seq, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_final_kernel, dim3(kReduceTerms), dim3(RF_BLOCK), 0, stream, partials, n_blocks, term_major ? 1 : 0,
                     sums_out, flags, seq);
  return hipGetLastError();
}

hipError_t launch_reduce_final_batch(const BatchPair* d_pairs, const BatchStep& step, int n_active, hipStream_t stream) {
  if (n_active <= 0) return hipSuccess;
  hipLaunchKernelGGL(reduce_final_batch_kernel, dim3(kReduceTerms, (unsigned)n_active), dim3(RF_BLOCK), 0, stream, d_pairs, step);
  return hipGetLastError();
}

hipError_t launch_transform(const float4* src, int n_s, const Xform& T, float4* out, hipStream_t stream) {
  if (n_s <= 0) return hipSuccess;
  int blocks = (n_s + 1023) / 1024;  // 4 points per lane
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(256), 0, stream, src, n_s, T, out);
  return hipGetLastError();
}

hipError_t launch_fill_keys(unsigned long long* keys, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n);
  return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void fingerprint_kernel(const float4* __restrict__ pts, int n, unsigned long long* __restrict__ acc) {
  unsigned long long s = 0ull;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    const unsigned long long w0 = ((unsigned long long)__float_as_uint(p.y) << 32) | __float_as_uint(p.x);
    const unsigned long long w1 = ((unsigned long long)__float_as_uint(p.w) << 32) | __float_as_uint(p.z);
    s += fp_point(w0, w1, (unsigned long long)i);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += (unsigned long long)__shfl_down((long long)s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(acc, s);  // integer addition: any order gives the same sum
}
// A result cloud straight into the context's pinned, mapped staging buffer (the caller's cloud is one host copy away), with its
// content fingerprint taken on the way: 16-byte stores, a wave writes 1 KiB of consecutive host addresses.  The number of points
// is on the device (the voxel filter's two counters) -- the host learns it from the same posted marker that tells it the points
// have arrived; nothing beyond `cap` points is written (the host falls back to the copy engine then).
__global__ __launch_bounds__(256) void publish_cloud_kernel(const float4* __restrict__ pts, const int* __restrict__ d_counts, int cap,
                                                            float4* __restrict__ host_out, unsigned long long* __restrict__ acc) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  int n = d_counts[0] + d_counts[1];
  n = n < 0 ? 0 : (n > cap ? cap : n);
  unsigned long long s = 0ull;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    v4f o;
    o.x = p.x; o.y = p.y; o.z = p.z; o.w = p.w;
    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(host_out + i));
    const unsigned long long w0 = ((unsigned long long)__float_as_uint(p.y) << 32) | __float_as_uint(p.x);
    const unsigned long long w1 = ((unsigned long long)__float_as_uint(p.w) << 32) | __float_as_uint(p.z);
    s += fp_point(w0, w1, (unsigned long long)i);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += (unsigned long long)__shfl_down((long long)s, off, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(acc, s);
}
}  // namespace

hipError_t launch_publish_cloud(const float4* pts, const int* d_counts, int n_most, int cap, float4* host_out, unsigned long long* d_acc,
                                hipStream_t stream) {
  if (n_most <= 0) return hipSuccess;
  int blocks = (n_most + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(publish_cloud_kernel, dim3(blocks), dim3(256), 0, stream, pts, d_counts, cap, host_out, d_acc);
  return hipGetLastError();
}

hipError_t launch_fingerprint(const float4* pts, int n, unsigned long long* d_acc, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_acc, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess || n <= 0) return e;
  int blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fingerprint_kernel, dim3(blocks), dim3(256), 0, stream, pts, n, d_acc);
  return hipGetLastError();
}

hipError_t launch_unpack_keys(const unsigned long long* keys, int n, int32_t* idx, float* d2, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n, idx, d2);
  return hipGetLastError();
}

}  // namespace icpgpu
