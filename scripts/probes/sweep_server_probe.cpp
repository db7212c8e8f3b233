// Dev probe (round 4): what would a RESIDENT point-to-point sweep kernel cost per ICP iteration beyond its work?
// P producer workgroups stay resident: per iteration they wait for a command (polling one of K 64-byte lines of fine-grained
// device memory that the host writes through the BAR), "work" for a given time, and publish N_PART x 17 partial sums as
// self-validating 16-byte granules; 17 reducer workgroups (one per term) poll the granules of their term, add them and store the
// sum into the host mailbox; the host polls the mailbox and writes the next command.  No kernel boundary, no launch.
// Prints the wall time per iteration for work = 0 (pure hand-over cost) and work = 60 us.
// Build: hipcc --offload-arch=gfx950 -O2 -o sweep_server_probe sweep_server_probe.cpp
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int TERMS = 17;
__host__ __device__ inline unsigned long long tag_of(unsigned long long seq, unsigned long long bits) {
  return (seq << 24) | ((bits ^ (bits >> 24) ^ (bits >> 48)) & 0xFFFFFFull);
}
__device__ __forceinline__ void gstore(unsigned long long* g, unsigned long long bits, unsigned long long seq_) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long seq = tag_of(seq_, bits);
  const u32x4 v = {(unsigned int)bits, (unsigned int)(bits >> 32), (unsigned int)seq, (unsigned int)(seq >> 32)};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 3" ::"v"(g), "v"(v) : "memory");
}
__device__ __forceinline__ bool gload(const unsigned long long* g, unsigned long long seq, unsigned long long& bits) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
  bits = ((unsigned long long)v.y << 32) | v.x;
  return (((unsigned long long)v.w << 32) | v.z) == seq;
}
__global__ __launch_bounds__(256) void server(int P, int n_part, int K, int iters, long long work_ticks, const unsigned int* cmd,
                                              unsigned long long* slots, unsigned long long* host_pairs, int* status) {
  const long long patience = 2000000;  // 20 ms
  __shared__ unsigned int s_seq;
  __shared__ double s_w[4];
  const int b = blockIdx.x;
  if (b < P) {
    for (int it = 1; it <= iters; ++it) {
      if (threadIdx.x < 64) {
        const unsigned int* line = cmd + (size_t)(b % K) * 16 + (threadIdx.x & 15);
        const long long t0 = (long long)wall_clock64();
        unsigned int got = 0xFFFFFFFFu;
        for (unsigned polls = 1;; ++polls) {
          const unsigned int w = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const unsigned int v = (unsigned int)__builtin_amdgcn_readlane((int)w, 12);
          if (v == (unsigned int)it || v == 0xFFFFFFFFu) { got = v; break; }
          if ((polls & 15u) == 0 && (long long)wall_clock64() - t0 > patience) break;
        }
        if (threadIdx.x == 0) s_seq = got;
      }
      __syncthreads();
      if (s_seq != (unsigned int)it) { if (threadIdx.x == 0 && s_seq != 0xFFFFFFFFu) *status = 1; return; }
      if (work_ticks > 0) {
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(8);
      }
      for (int u = b; u < n_part; u += P)
        if (threadIdx.x < TERMS) {
          const double v = (double)(u % 97) * 0.5 + (double)threadIdx.x;
          gstore(slots + 2 * ((size_t)(it & 1) * TERMS * n_part + (size_t)threadIdx.x * n_part + u), (unsigned long long)__double_as_longlong(v), (unsigned long long)it);
        }
      __syncthreads();
    }
  } else {
    const int term = b - P;
    for (int it = 1; it <= iters; ++it) {
      double v = 0.0;
      const unsigned long long* base = slots + 2 * ((size_t)(it & 1) * TERMS * n_part + (size_t)term * n_part);
      // all of a thread's granules in flight at once (one at a time: 9.2 us per iteration, the dependent reads dominating)
      constexpr int OWN = 13;  // ceil(3128 / 256)
      unsigned long long bits[OWN], tag[OWN];
      const long long t0 = (long long)wall_clock64();
      for (unsigned polls = 1;; ++polls) {
#pragma unroll
        for (int u = 0; u < OWN; ++u) {
          const int j = threadIdx.x + 256 * u;
          const unsigned long long* g = base + 2 * (size_t)(j < n_part ? j : threadIdx.x);
          bits[u] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          tag[u] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        bool all = true;
#pragma unroll
        for (int u = 0; u < OWN; ++u) all = all && tag[u] == tag_of((unsigned long long)it, bits[u]);
        if (all) break;
        if ((polls & 15u) == 0 && (long long)wall_clock64() - t0 > patience) { *status = 2; return; }
      }
#pragma unroll
      for (int u = 0; u < OWN; ++u)
        if (threadIdx.x + 256 * u < n_part) v += __longlong_as_double((long long)bits[u]);
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) {
        const double sum = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        gstore(host_pairs + 2 * term, (unsigned long long)__double_as_longlong(sum), (unsigned long long)it);
      }
      __syncthreads();
    }
  }
}
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int n_part = 3128;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned int* cmd;
  CK(hipExtMallocWithFlags((void**)&cmd, 4096, hipDeviceMallocFinegrained));
  CK(hipMemset(cmd, 0, 4096));
  unsigned long long* slots;
  const size_t slot_bytes = (size_t)2 * TERMS * n_part * 16;
  CK(hipExtMallocWithFlags((void**)&slots, slot_bytes, hipDeviceMallocFinegrained));
  CK(hipMemset(slots, 0, slot_bytes));
  unsigned long long* hp;
  CK(hipHostMalloc((void**)&hp, 4096, hipHostMallocMapped | hipHostMallocCoherent));
  unsigned long long* hp_dev;
  CK(hipHostGetDevicePointer((void**)&hp_dev, hp, 0));
  int* status;
  CK(hipHostMalloc((void**)&status, 64, hipHostMallocMapped | hipHostMallocCoherent));
  double expect[TERMS];
  for (int t = 0; t < TERMS; ++t) { expect[t] = 0; for (int u = 0; u < n_part; ++u) expect[t] += (double)(u % 97) * 0.5 + t; }
  const int configs[][3] = {{1984, 1, 0}, {1984, 64, 0}, {1984, 64, 6000}, {1024, 64, 0}, {1024, 64, 6000}, {256, 64, 0}};
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, server, 256, 0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int fit = per_cu * prop.multiProcessorCount;
  printf("%d workgroups of this kernel per CU, %d CUs: %d resident at most\n", per_cu, prop.multiProcessorCount, fit);
  for (const auto& cf : configs) {
    const int P = cf[0] + TERMS <= fit ? cf[0] : fit - TERMS - 8, K = cf[1];  // every workgroup must be resident: nobody waits for an undispatched one
    const long long work = cf[2];
    std::memset(hp, 0, 4096);
    *status = 0;
    CK(hipMemset(slots, 0, slot_bytes));
    CK(hipMemset(cmd, 0, 4096));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(server, dim3(P + TERMS), dim3(256), 0, s, P, n_part, K, iters, work, cmd, slots, hp_dev, status);
    volatile unsigned int* line = cmd;
    volatile unsigned long long* w = hp;
    bool failed = false;
    std::chrono::steady_clock::time_point t0;
    for (int it = 1; it <= iters && !failed; ++it) {
      if (it == 101) t0 = std::chrono::steady_clock::now();
      for (int l = 0; l < K; ++l) line[l * 16 + 12] = (unsigned int)it;
      _mm_sfence();
      const auto tw = std::chrono::steady_clock::now();
      for (unsigned spins = 1;; ++spins) {
        unsigned long long stale = 0;
        for (int t = 0; t < TERMS; ++t) stale |= (w[2 * t + 1] >> 24) ^ (unsigned long long)it;
        if (!stale) break;
        if ((spins & 0xFFFu) == 0 && (*status != 0 || std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count() > 200.0)) { failed = true; break; }
      }
      if (!failed)
        for (int t = 0; t < TERMS; ++t) {
          double v; unsigned long long bits = w[2 * t]; std::memcpy(&v, &bits, 8);
          if (v != expect[t]) { printf("iteration %d term %d: %g instead of %g\n", it, t, v, expect[t]); failed = true; break; }
        }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (iters - 100);
    for (int l = 0; l < K; ++l) line[l * 16 + 12] = 0xFFFFFFFFu;  // exit (whoever still waits)
    _mm_sfence();
    const hipError_t e = hipStreamSynchronize(s);
    printf("P %4d producers, K %2d command lines, work %3lld us: %s, status %d, %s%.2f us per iteration (hand-over = that minus the work)\n", P, K, work / 100,
           hipGetErrorString(e), *status, failed ? "FAILED " : "", us);
    fflush(stdout);
    if (e != hipSuccess) return 1;
  }
  return 0;
}
