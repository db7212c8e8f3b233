// waitvalue_probe.cpp -- can the launch of the NEXT sweep be taken off the ICP iteration's critical path?  (EXPERIMENTS.md section 10, open 1)
// A chain of dependent tiny kernels, each storing a tagged result into mapped host memory that the host polls:
//   A: host sees result k, THEN calls hipLaunchKernelGGL(k + 1)                          (what align_p2p does today)
//   B: [hipStreamWaitValue64(flag >= k + 1), kernel k + 1] are queued AHEAD; the host sees result k and writes the flag
//      (flag in signal memory / fine-grained device memory)
// Reported: microseconds per link.       hipcc --offload-arch=gfx950 -O2 waitvalue_probe.cpp -o waitvalue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>
__global__ void link(volatile unsigned long long* out, unsigned long long k, const float* arg) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((unsigned long long*)out, k + (unsigned long long)(arg ? arg[0] : 0.f), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 500, blocks = argc > 2 ? atoi(argv[2]) : 3128;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long* out; hipHostMalloc((void**)&out, 64, hipHostMallocMapped | hipHostMallocCoherent); *out = 0;
  unsigned long long* out_dev; hipHostGetDevicePointer((void**)&out_dev, out, 0);
  int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  // A
  for (int rep = 0; rep < 2; ++rep) {
    *out = 0; const double t0 = now();
    for (int k = 1; k <= n; ++k) {
      hipLaunchKernelGGL(link, dim3(blocks), dim3(256), 0, s, out_dev, (unsigned long long)k, (const float*)nullptr);
      while (*(volatile unsigned long long*)out != (unsigned long long)k) _mm_pause();
    }
    printf("A (launch after the result): %.2f us per link (%d blocks)\n", (now() - t0) / n, blocks);
  }
  if (!can) return 0;
  for (int kind = 0; kind < 2; ++kind) {
    unsigned long long* flag = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) : hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { printf("flag memory kind %d: %s\n", kind, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    unsigned long long* flag_host = flag;
    if (kind == 0) { hipMemset(flag, 0, 8); } else { *flag = 0; }
    for (int rep = 0; rep < 2; ++rep) {
      hipStreamSynchronize(s);
      if (kind == 0) hipMemset(flag, 0, 8); else *flag = 0;
      *out = 0;
      // the first link is launched plainly; link k + 1 is gated and queued before result k is awaited
      const double t0 = now();
      hipLaunchKernelGGL(link, dim3(blocks), dim3(256), 0, s, out_dev, 1ull, (const float*)nullptr);
      bool ok = true;
      for (int k = 1; k <= n; ++k) {
        if (k < n) {
          if (hipStreamWaitValue64(s, flag, (uint64_t)k, hipStreamWaitValueGte, ~0ull) != hipSuccess) { ok = false; break; }
          hipLaunchKernelGGL(link, dim3(blocks), dim3(256), 0, s, out_dev, (unsigned long long)(k + 1), (const float*)nullptr);
        }
        while (*(volatile unsigned long long*)out != (unsigned long long)k) _mm_pause();
        // "host solve", then open the gate of link k + 1
        if (kind == 0) { hipStreamWriteValue64(0, flag, (uint64_t)k, 0); /* via a second queue: not what we want */ }
        else { *(volatile unsigned long long*)flag_host = (unsigned long long)k; _mm_sfence(); }
      }
      if (!ok) { printf("hipStreamWaitValue64 failed: %s\n", hipGetErrorString(hipGetLastError())); break; }
      printf("B (%s flag, gated launch queued ahead): %.2f us per link\n", kind == 0 ? "signal-memory (written by hipStreamWriteValue64 on the null stream)" : "mapped host memory", (now() - t0) / n);
    }
  }
  return 0;
}
