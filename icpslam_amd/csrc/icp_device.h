// icp_device.h -- device-side helpers shared by the kernel translation units (internal).
// The arithmetic contract of DESIGN.md section 3 lives here: every kernel transforms and measures through these.
#pragma once

#include <hip/hip_runtime.h>

#include "icp_kernels.h"

namespace icpgpu {

__device__ __forceinline__ void xform_point(const Xform& T, float x, float y, float z, float& px, float& py, float& pz) {
  px = __builtin_fmaf(T.m[2], z, __builtin_fmaf(T.m[1], y, __builtin_fmaf(T.m[0], x, T.m[3])));
  py = __builtin_fmaf(T.m[6], z, __builtin_fmaf(T.m[5], y, __builtin_fmaf(T.m[4], x, T.m[7])));
  pz = __builtin_fmaf(T.m[10], z, __builtin_fmaf(T.m[9], y, __builtin_fmaf(T.m[8], x, T.m[11])));
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, float px, float py, float pz) {
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// {a, b} as ONE 16-byte store written through to system memory (sc0 sc1): a reader never sees half a pair (observed on gfx950,
// not architecturally promised: the tags the callers put into b are self-validating where it matters).
//
// The s_nop behind the store is REQUIRED.  gfx950 reads the data (and address) VGPRs of a store wider than 64 bits a few cycles
// after issue; the compiler inserts the wait states for its own stores, but an inline-asm store is opaque to its hazard
// recogniser, and the very next VALU instruction may overwrite those registers.  Measured (round 4, scripts/probes/
// granule_probe.cpp): two such stores in a row, the second one's value computed between them -- lane 12's low 8 bytes of the
// FIRST store arrive stale on every round; with four wait states behind each store: never, in 10^6 granules.
__device__ __forceinline__ void store_pair_system(void* p, unsigned long long a, unsigned long long b) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {(unsigned int)a, (unsigned int)(a >> 32), (unsigned int)b, (unsigned int)(b >> 32)};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
}

// A result pair {value bits, tag(number, bits)} (icp_kernels.h: mailbox_tag) into host-visible memory.  `number` may carry
// kMailboxReleaseBit: the classic form then -- value, system-scope release, tag -- instead of the single 16-byte store.
__device__ __forceinline__ void store_result_pair(void* p, unsigned long long bits, unsigned long long number) {
  const unsigned long long tag = mailbox_tag(number & ~kMailboxReleaseBit, bits);
  if (number & kMailboxReleaseBit) {
    unsigned long long* w = static_cast<unsigned long long*>(p);
    __hip_atomic_store(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    store_pair_system(p, bits, tag);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}


// 17-term accumulation of one accepted pair (p = transformed source, q = target), double precision.
__device__ __forceinline__ void accumulate_pair(double (&acc)[kReduceTerms], float px, float py, float pz, float qx,
                                                float qy, float qz, float d2) {
  const double p[3] = {(double)px, (double)py, (double)pz};
  const double q[3] = {(double)qx, (double)qy, (double)qz};
  acc[0] += 1.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    acc[1 + a] += p[a];
    acc[4 + a] += q[a];
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[7 + 3 * a + b] += q[a] * p[b];
  }
  acc[16] += (double)d2;
}

// Block-level fixed-order reduction of the per-lane accumulators into partials[block][17]. WAVES = blockDim.x / 64.
template <int WAVES>
__device__ __forceinline__ void block_reduce_store(const double (&acc)[kReduceTerms], double* __restrict__ partials) {
  __shared__ double wsum[WAVES][kReduceTerms];
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
    for (int w = 0; w < WAVES; ++w) v += wsum[w][threadIdx.x];
    partials[(size_t)blockIdx.x * kReduceTerms + threadIdx.x] = v;
  }
}

}  // namespace icpgpu
