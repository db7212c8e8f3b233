// Probe (round 5): what the pieces of an icpgpu context cost to create (icpgpu_align_batch creates up to 64 worker contexts on its
// first call).  hipcc --offload-arch=gfx950 -O2 scripts/probes/create_probe.cpp -o scripts/probes/create_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  (void)hipSetDevice(0);
  void* w;
  (void)hipMalloc(&w, 256);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  printf("hipStreamCreateWithFlags, streams 1..24 [us]:");
  for (int k = 0; k < 24; ++k) {
    auto t0 = now();
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    printf(" %.0f", us(t0, now()));
  }
  printf("\n");
  for (int rep = 0; rep < 2; ++rep) {
    auto t1 = now(); hipEvent_t ev[4]; for (auto& e : ev) (void)hipEventCreate(&e);
    auto t2 = now(); void* h1; (void)hipHostMalloc(&h1, 512, hipHostMallocMapped | hipHostMallocCoherent);
    auto t3 = now(); void* d1; (void)hipHostGetDevicePointer(&d1, h1, 0);
    auto t4 = now(); void* h2; (void)hipHostMalloc(&h2, 64, hipHostMallocDefault);
    auto t5 = now(); void* p1; (void)hipMalloc(&p1, 1024 * 17 * 8);
    auto t6 = now(); void* p2; (void)hipMalloc(&p2, 17 * 8);
    auto t7 = now(); void* fg; (void)hipExtMallocWithFlags(&fg, 4096, hipDeviceMallocFinegrained);
    auto t8 = now(); (void)hipMemset(fg, 0, 4096);
    auto t9 = now(); hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    auto t10 = now();
    printf("4 events %.0f | hostmalloc coherent %.0f | getdevptr %.0f | hostmalloc default %.0f | hipMalloc 139KB %.0f | hipMalloc 136B %.0f | finegrained %.0f | memset %.0f | device properties %.0f us\n",
           us(t1,t2), us(t2,t3), us(t3,t4), us(t4,t5), us(t5,t6), us(t6,t7), us(t7,t8), us(t8,t9), us(t9,t10));
  }
  return 0;
}
