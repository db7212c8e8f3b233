// granule_probe.cpp -- probe for gicp_solve_kernel's all-gather: B workgroups publish 28 self-validating 16-byte granules per round
// into double-buffered FINE-GRAINED device memory (one sc0 sc1 store each), every workgroup's wave 0 gathers all of them.
// Reports the time per round and any granule that never became valid.   hipcc --offload-arch=gfx950 -O3 granule_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int G = 28;
__device__ __forceinline__ unsigned long long tag_of(unsigned long long seq, unsigned long long bits) {
  return (seq << 24) | ((bits ^ (bits >> 24) ^ (bits >> 48)) & 0xFFFFFFull);
}
template <bool NOPS = false>
__device__ __forceinline__ void gstore(unsigned long long* g, unsigned long long bits, unsigned long long seq) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long tag = tag_of(seq, bits);
  const u32x4 v = {(unsigned int)bits, (unsigned int)(bits >> 32), (unsigned int)tag, (unsigned int)(tag >> 32)};
  if (NOPS) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 3" ::"v"(g), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(g), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ bool gload(const unsigned long long* g, unsigned long long seq, unsigned long long& bits) {
  unsigned long long tag;
  if (MODE == 0) {
    bits = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    tag = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
    bits = ((unsigned long long)v.y << 32) | v.x;
    tag = ((unsigned long long)v.w << 32) | v.z;
  }
  return tag == tag_of(seq, bits);
}
// STRIDE: words per granule (2 = packed 16-byte granules, 4 = one granule per 32-byte sector).  SPLIT: the 28 granules leave in two
// store instructions (lanes 0..13: granules 0..13, then 14..27), as gicp_solve_kernel's first version did: granules 13 and 14
// then share a 32-byte sector but come from different instructions.
template <int MODE, int STRIDE, bool SPLIT, bool NOPS = false>
__global__ __launch_bounds__(256) void probe(unsigned long long* slots, int rounds, unsigned long long seq0, unsigned long long* report) {
  const int B = gridDim.x;
  unsigned long long bad = 0, t_first = 0, t_last = 0;
  for (int r = 1; r <= rounds; ++r) {
    const unsigned long long seq = seq0 + r;
    unsigned long long* parity = slots + (size_t)(r & 1) * B * G * STRIDE;
    if (!SPLIT) {
      if (threadIdx.x < G) gstore(parity + (size_t)blockIdx.x * G * STRIDE + STRIDE * threadIdx.x, (seq * 1000003ull + blockIdx.x * 131ull + threadIdx.x) * 0x9E3779B97F4A7C15ull, seq);
    } else if (threadIdx.x < G / 2) {
      gstore<NOPS>(parity + (size_t)blockIdx.x * G * STRIDE + STRIDE * threadIdx.x, (seq * 1000003ull + blockIdx.x * 131ull + threadIdx.x) * 0x9E3779B97F4A7C15ull, seq);
      gstore<NOPS>(parity + (size_t)blockIdx.x * G * STRIDE + STRIDE * (threadIdx.x + G / 2), (seq * 1000003ull + blockIdx.x * 131ull + threadIdx.x + G / 2) * 0x9E3779B97F4A7C15ull, seq);
    }
    if (threadIdx.x < 64) {
      const int q = threadIdx.x >> 2, p = threadIdx.x & 3;
      if (q < 14) {
        for (int b = p; b < B; b += 4) {
          for (int half = 0; half < 2; ++half) {
            const int gi = q + 14 * half;
            const unsigned long long* g = parity + (size_t)b * G * STRIDE + STRIDE * gi;
            unsigned long long bits = 0;
            const long long t0 = (long long)wall_clock64();
            bool ok = false;
            while (!(ok = gload<MODE>(g, seq, bits)))
              if ((long long)wall_clock64() - t0 > 20000) break;
            if (!ok || bits != (seq * 1000003ull + (unsigned long long)b * 131ull + gi) * 0x9E3779B97F4A7C15ull) {
              if (!bad && atomicCAS(&report[7], 0ull, 1ull) == 0ull) { report[8 + 0] = r; report[8 + 1] = b; report[8 + 2] = gi; report[8 + 3] = bits; report[8 + 4] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); report[8 + 5] = blockIdx.x; report[8 + 6] = seq; report[8+7] = ok; }
              ++bad;
            }
          }
        }
      }
    }
    __syncthreads();
    if (r == 1) t_first = wall_clock64();
    t_last = wall_clock64();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&report[0], bad);
    if (blockIdx.x == 0) { report[1] = t_first; report[2] = t_last; }
  }
  unsigned long long any = bad;
  for (int d = 32; d >= 1; d >>= 1) any += __shfl_xor(any, d, 64);
  if ((threadIdx.x & 63) == 0 && any) atomicAdd(&report[3], any);
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long *slots, *report;
  hipExtMallocWithFlags((void**)&slots, 2 * 256 * G * 32, hipDeviceMallocFinegrained);
  hipMemset(slots, 0, 2 * 256 * G * 32);
  hipHostMalloc((void**)&report, 256, hipHostMallocMapped);
  unsigned long long seq0 = 8192;
  for (int mode = 0; mode < 5; ++mode)
    for (int B : {4, 20, 23, 64}) {
      for (int i = 0; i < 32; ++i) report[i] = 0;
      if (mode == 0) hipLaunchKernelGGL((probe<0, 2, false>), dim3(B), dim3(256), 0, 0, slots, rounds, seq0, report);
      else if (mode == 1) hipLaunchKernelGGL((probe<1, 2, false>), dim3(B), dim3(256), 0, 0, slots, rounds, seq0, report);
      else if (mode == 2) hipLaunchKernelGGL((probe<0, 2, true>), dim3(B), dim3(256), 0, 0, slots, rounds, seq0, report);
      else if (mode == 3) hipLaunchKernelGGL((probe<0, 4, true>), dim3(B), dim3(256), 0, 0, slots, rounds, seq0, report);
      else hipLaunchKernelGGL((probe<0, 2, true, true>), dim3(B), dim3(256), 0, 0, slots, rounds, seq0, report);
      hipError_t e = hipDeviceSynchronize();
      seq0 += rounds + 8192;
      static const char* names[] = {"one store, 2 x 8-byte atomic loads", "one store, 16-byte loads", "TWO stores sharing a sector, packed granules", "TWO stores, one granule per 32-byte sector", "TWO stores, packed, s_nop 3 behind each store"};
      printf("%-48s B %3d: %s, bad granules %llu (wave total %llu), %.3f us per round", names[mode], B, hipGetErrorString(e), report[0], report[3],
             (double)(report[2] - report[1]) * 0.01 / (rounds - 1));
      if (report[3]) printf("  first: round %llu block %llu granule %llu bits %llx tag number %llu checksum %llx (expected number %llu; valid %llu) seen by block %llu", report[8], report[9], report[10], report[11], report[12] >> 24, report[12] & 0xFFFFFF, report[14], report[15], report[13]);
      printf("\n"); fflush(stdout);
    }
  return 0;
}
