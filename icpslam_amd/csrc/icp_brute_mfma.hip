// icp_brute_mfma.hip -- brute-force exact nearest neighbour on the MATRIX cores (row a2 of SURVEY.md section 8(a);
// north_star's LDS-tiled brute force, the arithmetic of PCL's CorrespondenceEstimation::determineCorrespondences reached from
// /root/reference/src/icpslam/icp_odometer.cpp:198 and src/icpslam/octree_mapper.cpp:114).
//
// Why: the plain-VALU kernel (icp_kernels.hip) needs 7 vector instructions per (source, target) pair and sits at 79 % of
// what the vector ALUs can issue -- 62 TFLOP/s by the 8 flop / pair convention, 40 % of the chip's f32 peak, because that
// peak (157.3 TFLOP/s) is only reachable with packed or matrix f32 (v_pk_fma_f32 issues at half rate on gfx950: the packed
// variant of the VALU kernel measured +-3 %).  The f32 MFMA runs at the full 157.3 TFLOP/s and is bit-for-bit an fmaf
// chain (MI355X guide), but the contract's distance |q - p|^2 = fma(dz, dz, fma(dy, dy, dx * dx)) is a function of
// DIFFERENCES, not a contraction over (source, target) -- so the matrix cores cannot produce it.  They can produce a
// certified LOWER BOUND of it for 32 x 32 pairs at a time, and nearly every pair is settled by that bound alone:
//
//   centre c (one per workgroup: its 256 sources are neighbours in space, they come in cell order), p' = p - c, q' = q - c
//   |q - p|^2 - |p'|^2  =  |q'|^2 - 2 p'.q'                      (exact in real arithmetic)
//   s_ij = fma(1, |q'_j|^2 - tau_j, fma(-2p'z, q'z, fma(-2p'y, q'y, (-2p'x) q'x)))      two v_mfma_f32_32x32x2_f32, K = 4
//   tau_j = 2^-18 (P^2 + |q'_j|^2),  P = max |p'| of the workgroup  >= every rounding on the way (see below)
//   =>  s_ij <= |q_j - p_i|^2 - |p'_i|^2   for every pair.
//
// Each lane keeps, per source, the best EXACT key found so far (contract arithmetic on the original coordinates, lowest
// index among ties) and the bound B = its distance - |p'|^2 (+ margin).  A target can only beat or tie the best if
// s_ij <= B; the 16 values a lane receives from a 32 x 32 tile are folded with v_min3 and compared ONCE; only a lane that
// sees s <= B evaluates those targets exactly and tightens its bound.  For targets in arbitrary order the bound improves
// ~ln(N) times per source, so the exact path runs for ~20 of 200 000 targets; everything else costs two MFMAs per 1024
// pairs plus ~20 vector instructions per tile.  The result is the exact key -- same bits as the VALU kernel and the oracle.
//
// Error budget of the lower bound (all in units of 2^-24 = one float rounding, relative to the magnitudes named):
//   p' and q' are rounded differences (1 each, of |p'|, |q'|): moves |q' - p'|^2 by <= 2 |q-p| (|p'| + |q'|) 2^-24 sqrt(3)
//   |q'|^2 by three roundings (3 of |q'|^2); the four fused steps of the chain (4 of 2|p'||q'| + |q'|^2)
//   sum <= 2^-24 * 16 (P + |q'|)^2 <= 2^-24 * 32 (P^2 + |q'|^2) = 2^-19 (P^2 + |q'|^2); tau carries a further factor 2.
// tests/test_gpu_parity.py / test_gpu_grid.py / test_gpu_fullsize.py compare this kernel's keys with the VALU kernel and the
// oracle bit for bit (ties, duplicates, NaN / inf points, far outliers, 200k x 200k and 200k x 1M).
#include <hip/hip_runtime.h>

#include "icp_env.h"

#include <cstdio>
#include <math.h>

#include <cstdlib>

#include "icp_device.h"
#include "icp_kernels.h"

namespace icpgpu {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MF_BLOCK = 256;   // 4 waves
constexpr int MF_TILE = 1024;   // target points per LDS tile (16 KiB), shared by the 4 waves
constexpr float kTau = 3.814697265625e-06f;  // 2^-18

__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// G groups of 32 sources per wave (both half-waves hold the same 32 sources; each sees 16 of a tile's 32 targets)
template <int G, bool PIPE>
__device__ __forceinline__ void nn_brute_mfma_body(const float4* __restrict__ src_sorted, int n_q,
                                                   const float4* __restrict__ tgt, int n_t, const Xform& T,
                                                   int tgt_per_split, int splits, unsigned long long* __restrict__ keys,
                                                   const unsigned long long* __restrict__ seed, int debug_no_exact) {
  // a target tile in LDS, twice: the raw points (for the exact evaluations) and the MFMA's A operands, ready to use --
  // (q'x, q'y) and (q'z, |q'|^2 - tau) -- computed ONCE per tile and workgroup instead of per wave and step (16 vector
  // instructions per step less; measured effect on the sweep: within noise -- the kernel sits at 1.55-1.6x the time its MFMAs
  // alone would take whatever else is trimmed, profiles/r02_brute_force_mfma.txt).
  __shared__ float4 tile[MF_TILE];
  __shared__ float2 a01s[MF_TILE], a23s[MF_TILE];
  __shared__ float s_centre[4];
  __shared__ float s_pmax[MF_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int q0 = (blockIdx.x * (MF_BLOCK / 64) + wave) * (32 * G);

  // sources of this wave: group g, column col (transformed: the contract's p = T * s)
  float px[G], py[G], pz[G];
  int orig[G];
  bool valid[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k = q0 + g * 32 + col;
    valid[g] = k < n_q;
    const float4 s = src_sorted[min(k, n_q - 1)];
    xform_point(T, s.x, s.y, s.z, px[g], py[g], pz[g]);
    orig[g] = __float_as_int(s.w);
  }
  // the workgroup's centre: its first source (sources arrive in cell order: the other 32 G x 4 - 1 are close by)
  if (threadIdx.x == 0) {
    s_centre[0] = px[0];
    s_centre[1] = py[0];
    s_centre[2] = pz[0];
  }
  __syncthreads();
  const float cx = s_centre[0], cy = s_centre[1], cz = s_centre[2];
  // B operands (K x N = sources): lane supplies B[k = half][n = col] -- first MFMA k = 0, 1 = (-2p'x, -2p'y), second
  // k = 2, 3 = (-2p'z, 1)
  float b01[G], b23[G], p2[G], pmax2 = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float ux = px[g] - cx, uy = py[g] - cy, uz = pz[g] - cz;
    b01[g] = half ? -2.0f * uy : -2.0f * ux;
    b23[g] = half ? 1.0f : -2.0f * uz;
    p2[g] = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    pmax2 = fmaxf(pmax2, p2[g]);  // (a NaN source never matches anything: fmaxf drops it here, its s_ij are NaN below)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pmax2 = fmaxf(pmax2, __shfl_xor(pmax2, off, 64));
  if (lane == 0) s_pmax[wave] = pmax2;
  __syncthreads();
  pmax2 = fmaxf(fmaxf(s_pmax[0], s_pmax[1]), fmaxf(s_pmax[2], s_pmax[3]));  // P^2 of the WORKGROUP: tau is shared by its waves

  unsigned long long best[G];
  float bound[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    best[g] = kEmptyKey;
    bound[g] = INFINITY;
    // Seed (optional): the neighbour this source found in the previous sweep over the same target.  It is a target point
    // whatever the transform is now, so its exact key is a valid candidate and its distance a valid bound from the first
    // tile on: in an ICP iteration nearly nothing undercuts it (without a seed the bound improves ~ln(N) times per source).
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
  // target tiles: the NEXT tile's global loads are in flight (in registers) while this one is being worked on
  float4 nxt[MF_TILE / MF_BLOCK];
#pragma unroll
  for (int u = 0; u < MF_TILE / MF_BLOCK; ++u) nxt[u] = tgt[min(j0 + u * MF_BLOCK + (int)threadIdx.x, n_t - 1)];
  for (int jt = j0; jt < j1; jt += MF_TILE) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MF_TILE / MF_BLOCK; ++u) {
      const int k = u * MF_BLOCK + threadIdx.x;
      const float4 q = nxt[u];
      tile[k] = q;
      // A operands (M x K = targets): row k supplies (q'x, q'y | q'z, |q'|^2 - tau); a row past the end can never undercut a bound
      const float vx = q.x - cx, vy = q.y - cy, vz = q.z - cz;
      const float q2 = __builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx));
      const bool real = jt + k < j1;
      a01s[k] = real ? make_float2(vx, vy) : make_float2(0.f, 0.f);
      a23s[k] = real ? make_float2(vz, q2 - (pmax2 + q2) * kTau) : make_float2(0.f, INFINITY);
    }
    __syncthreads();
    if (jt + MF_TILE < j1) {
#pragma unroll
      for (int u = 0; u < MF_TILE / MF_BLOCK; ++u) nxt[u] = tgt[min(jt + MF_TILE + u * MF_BLOCK + (int)threadIdx.x, n_t - 1)];
    }
    const int lim = min(MF_TILE, j1 - jt);
    const float* __restrict__ a01p = reinterpret_cast<const float*>(a01s) + half;
    const float* __restrict__ a23p = reinterpret_cast<const float*>(a23s) + half;
    // One step = 32 targets against the wave's 32 G sources: 2 G MFMAs, then the fold of their 16 G values per lane.  The
    // steps are SOFTWARE-PIPELINED (PIPE): the MFMAs of step k + 1 are issued into a second accumulator set before step k's
    // values are folded, so the ~27 vector instructions of a fold run under the wave's OWN matrix work -- round 2's counters
    // showed matrix and vector time adding up (64 % + 27 %) instead of overlapping when a wave folds right behind its MFMAs
    // and leaves the overlap to the other waves of the SIMD.
    auto issue = [&](floatx16 (&acc)[G], int st) {
      const float a01 = a01p[2 * (st + col)], a23 = a23p[2 * (st + col)];  // lane supplies A[m = col][k = half]
      // all the groups' first MFMAs, then the second ones (each depends on its first: interleaved, neither waits)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01[g], zero, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a23, b23[g], acc[g], 0, 0, 0);
    };
    auto fold = [&](floatx16 (&acc)[G], int st) {
      // The folds below are v_min3_f32 by hand (fminf on MFMA outputs makes the compiler canonicalise every operand first:
      // 6 extra instructions per fold).  Inline assembly is opaque to the compiler's hazard recogniser, which is what
      // inserts the wait states between an MFMA and the first vector read of its result -- so every accumulator is first
      // read by an instruction the compiler CAN see (it waits there), and nothing may be scheduled across.
      __builtin_amdgcn_sched_barrier(0);
      int touched = 0;
#pragma unroll
      for (int g = 0; g < G; ++g) touched |= __builtin_amdgcn_readfirstlane(__float_as_int(acc[g][15]));
      asm volatile("" ::"s"(touched));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const floatx16 d = acc[g];
        // d[v]: target row 8 * (v / 4) + 4 * half + (v % 4) of this step, source column col.  A NaN (a non-finite point) is
        // dropped by the minimum and fails the comparison below.
        const float mn = min3f(min3f(min3f(d[0], d[1], d[2]), min3f(d[3], d[4], d[5]), min3f(d[6], d[7], d[8])),
                               min3f(d[9], d[10], d[11]), min3f(min3f(d[12], d[13], d[14]), d[15], d[15]));
        if (!debug_no_exact && __ballot(mn <= bound[g])) {  // rare: some lane's bound is undercut -- those targets are evaluated exactly
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
                  // everything that can still beat or tie it has  |q - p|^2 - |p'|^2 <= e - |p'|^2: the bound, widened
                  // for the rounding of p2 and of this subtraction
                  bound[g] = __builtin_fmaf(e + p2[g], 9.5367431640625e-07f, e - p2[g]);
                }
              }
            }
          }
        }
      }
    };
    if (PIPE) {
      floatx16 acc_a[G], acc_b[G];
      issue(acc_a, 0);
      for (int st = 0;; st += 64) {
        const bool more_b = st + 32 < lim;
        if (more_b) issue(acc_b, st + 32);
        fold(acc_a, st);
        if (!more_b) break;
        const bool more_a = st + 64 < lim;
        if (more_a) issue(acc_a, st + 64);
        fold(acc_b, st + 32);
        if (!more_a) break;
      }
    } else {
      for (int st = 0; st < lim; st += 32) {
        floatx16 acc[G];
        issue(acc, st);
        fold(acc, st);
      }
    }
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

template <int G, bool PIPE>
__global__ __launch_bounds__(MF_BLOCK) void nn_brute_mfma_kernel(const float4* __restrict__ src_sorted, int n_q,
                                                                 const float4* __restrict__ tgt, int n_t, Xform T,
                                                                 int tgt_per_split, int splits,
                                                                 unsigned long long* __restrict__ keys,
                                                                 const unsigned long long* __restrict__ seed, int debug_no_exact) {
  nn_brute_mfma_body<G, PIPE>(src_sorted, n_q, tgt, n_t, T, tgt_per_split, splits, keys, seed, debug_no_exact);
}

// the pipelined body held to 128 registers (four waves per SIMD instead of three; a few spills in return) -- ICPGPU_MFMA_PIPE=2
template <int G>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void nn_brute_mfma_capped_kernel(
    const float4* __restrict__ src_sorted, int n_q, const float4* __restrict__ tgt, int n_t, Xform T, int tgt_per_split,
    int splits, unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ seed, int debug_no_exact) {
  nn_brute_mfma_body<G, true>(src_sorted, n_q, tgt, n_t, T, tgt_per_split, splits, keys, seed, debug_no_exact);
}

}  // namespace

// src_sorted: the source in cell order with the ORIGINAL index in .w (what the grid machinery's sorted copy holds); keys are
// written at the original indices and must be pre-filled with kEmptyKey (points missing from src_sorted -- the non-finite ones
// -- stay unmatched, and target splits merge by atomic min).
// seed (optional, n_s keys indexed like `keys`): every source's neighbour from an earlier sweep over the SAME target array.
hipError_t launch_nn_brute_mfma(const float4* src_sorted, int n_q, const float4* tgt, int n_t, const Xform& T, int num_cus,
                                unsigned long long* keys, const unsigned long long* seed, hipStream_t stream) {
  if (n_q <= 0 || n_t <= 0) return hipSuccess;
  // experiments: ICPGPU_MFMA_G (sources per wave / 32: 2 or 4), ICPGPU_MFMA_WAVES (target waves per SIMD the splits aim at),
  // ICPGPU_MFMA_NO_EXACT (timing only: the exact path is skipped, results are wrong)
  static const int g_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_MFMA_G"); return e ? atoi(e) : 2; }();
  static const int waves_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_MFMA_WAVES"); return e ? atoi(e) : 32; }();
  static const int no_exact = [] {
    if (!ICPGPU_DEV_ENV("ICPGPU_MFMA_NO_EXACT")) return 0;
    fprintf(stderr, "[icpgpu] WARNING: ICPGPU_MFMA_NO_EXACT is set -- the matrix-core search skips its exact path, every brute-force result "
                    "of this process is WRONG (a timing experiment's switch, never a production setting)\n");
    return 1;
  }();
  const int G = g_env == 4 ? 4 : 2;
  const int per_block = (MF_BLOCK / 64) * 32 * G;
  const int grid_x = (n_q + per_block - 1) / per_block;
  // target splits for ~32 waves' worth of workgroups per SIMD (measured at 200k x 200k: 2 -> 4.8 ms, 8 -> 3.45, 16 -> 3.23,
  // 32 -> 3.14: the tail of the last round of workgroups), never less than one tile per split
  int splits = (num_cus * waves_env + grid_x - 1) / grid_x;
  const int max_splits = (n_t + MF_TILE - 1) / MF_TILE;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int per = (n_t + splits - 1) / splits;
  per = ((per + MF_TILE - 1) / MF_TILE) * MF_TILE;
  splits = (n_t + per - 1) / per;
  static const int pipe_env = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_MFMA_PIPE"); return e ? atoi(e) : 0; }();
#define ICPGPU_MFMA_LAUNCH(...)                                                                                         \
  hipLaunchKernelGGL((__VA_ARGS__), dim3(grid_x, splits), dim3(MF_BLOCK), 0, stream, src_sorted, n_q, tgt, n_t, T, per, splits, \
                     keys, seed, no_exact)
  if (G == 4) {
    if (pipe_env) ICPGPU_MFMA_LAUNCH(nn_brute_mfma_kernel<4, true>);
    else ICPGPU_MFMA_LAUNCH(nn_brute_mfma_kernel<4, false>);
  } else {
    if (pipe_env == 2) ICPGPU_MFMA_LAUNCH(nn_brute_mfma_capped_kernel<2>);
    else if (pipe_env) ICPGPU_MFMA_LAUNCH(nn_brute_mfma_kernel<2, true>);
    else ICPGPU_MFMA_LAUNCH(nn_brute_mfma_kernel<2, false>);
  }
#undef ICPGPU_MFMA_LAUNCH
  return hipGetLastError();
}

}  // namespace icpgpu
