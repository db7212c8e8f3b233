// Dev probe (round 4): what a small device -> host hand-over costs between two dependent kernels of one stream:
//   (a) hipMemcpyAsync (24 B, pinned destination) + hipStreamSynchronize   (what the grid builds did until round 4)
//   (b) a one-wave "post" kernel that stores {value, tag} pairs into mapped host memory (sc0 sc1) while the host polls
// Build: hipcc --offload-arch=gfx950 -O2 -o roundtrip_probe roundtrip_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work_kernel(int* d, int v) { if (threadIdx.x < 6) atomicMax(&d[threadIdx.x], v + (int)threadIdx.x); }
__global__ void post_kernel(const int* d, int n, unsigned long long* host_pairs, unsigned long long seq) {
  if ((int)threadIdx.x < n) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long a = (unsigned long long)(unsigned int)d[threadIdx.x], b = seq;
    const u32x4 v = {(unsigned int)a, (unsigned int)(a >> 32), (unsigned int)b, (unsigned int)(b >> 32)};
    void* p = host_pairs + 2 * threadIdx.x;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 3" ::"v"(p), "v"(v) : "memory");
  }
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
  int* h; CK(hipHostMalloc(&h, 64, hipHostMallocDefault));
  unsigned long long* hp; CK(hipHostMalloc(&hp, 256, hipHostMallocMapped | hipHostMallocCoherent));
  unsigned long long* hp_dev; CK(hipHostGetDevicePointer((void**)&hp_dev, hp, 0));
  std::memset(hp, 0, 256);
  const int reps = 2000;
  int tick = 0;
  for (int mode = 0; mode < 3; ++mode) {
    for (int warm = 0; warm < 2; ++warm) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 1; r <= reps; ++r) {
        ++tick;
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, s, d, tick);
        if (mode == 0) {
          CK(hipMemcpyAsync(h, d, 24, hipMemcpyDeviceToHost, s));
          CK(hipStreamSynchronize(s));
          if (h[0] != tick) { printf("wrong value\n"); return 1; }
        } else if (mode == 1) {
          const unsigned long long seq = (unsigned long long)tick;
          hipLaunchKernelGGL(post_kernel, dim3(1), dim3(64), 0, s, d, 6, hp_dev, seq);
          volatile unsigned long long* w = hp;
          for (;;) { bool all = true; for (int k = 0; k < 6; ++k) all = all && w[2 * k + 1] == seq; if (all) break; }
          if ((int)w[0] != tick) { printf("wrong value (post)\n"); return 1; }
        } else {
          CK(hipStreamSynchronize(s));  // the launch + an empty synchronisation alone
        }
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      if (warm) printf("%s: %.2f us per round trip\n", mode == 0 ? "kernel + hipMemcpyAsync(24 B D2H) + hipStreamSynchronize" : mode == 1 ? "kernel + post kernel (pairs into mapped host memory) + host poll" : "kernel + hipStreamSynchronize (no copy)", us);
    }
  }
  return 0;
}
