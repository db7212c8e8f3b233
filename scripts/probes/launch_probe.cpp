// Dev tool: host time of one kernel launch call on this stack -- hipLaunchKernelGGL with no / few / many (200 B) arguments,
// hipModuleLaunchKernel with a pre-packed argument buffer -- on a non-blocking stream, queue never empty vs. always empty.
// hipcc --offload-arch=gfx950 -O2 -o launch_probe launch_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct X12 { float m[12]; };
struct G14 { int a[8]; float b[6]; };
__global__ void k_empty() {}
__global__ void k_few(int* p, int a) { if (a == 12345) *p = a; }
__global__ void k_many(const float4* a, int b, int c, int d, X12 T, const float4* e, const int* f, G14 g, float h, unsigned long long* i,
                       double* j, int* k, int* l, float4* m, int n, int o, unsigned long long* q) {
  if (b == 12345) *l = (int)T.m[3] + g.a[2] + c + d + n + o + (int)h + (a && e && f && i && j && k && m && q);
}
template <typename F> static double time_calls(F&& f, hipStream_t s, int n, bool drain) {
  double tot = 0;
  for (int i = 0; i < n; ++i) {
    if (drain) (void)hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    f();
    tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  }
  (void)hipStreamSynchronize(s);
  return tot / n;
}
int main() {
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d;
  (void)hipMalloc(&d, 64);
  X12 T{};
  G14 g{};
  auto many = [&] { hipLaunchKernelGGL(k_many, dim3(1), dim3(64), 0, s, nullptr, 1, 2, 3, T, nullptr, nullptr, g, 1.f, nullptr, nullptr, d, d, nullptr, 4, 5, nullptr); };
  auto few = [&] { hipLaunchKernelGGL(k_few, dim3(1), dim3(64), 0, s, d, 1); };
  auto empty = [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); };
  hipFunction_t fn = nullptr;
  hipError_t e = hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(k_many));
  struct Packed {
    const float4* a; int b, c, d; X12 T; const float4* e; const int* f; G14 g; float h; unsigned long long* i; double* j; int* k; int* l;
    float4* m; int n, o; unsigned long long* q;
  } pk{};
  pk.b = 1; pk.k = d; pk.l = d;
  size_t sz = sizeof(pk);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  auto module = [&] { (void)hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, s, nullptr, extra); };
  for (int w = 0; w < 200; ++w) { many(); few(); empty(); if (fn) module(); }
  (void)hipStreamSynchronize(s);
  std::printf("hipGetFuncBySymbol: %s; sizeof(Packed) = %zu\n", hipGetErrorString(e), sz);
  for (int drain = 0; drain < 2; ++drain) {
    std::printf("%s: empty %.2f us | 2 args %.2f us | 17 args (200 B) %.2f us", drain ? "stream idle before every call" : "back to back              ",
                time_calls(empty, s, 2000, drain), time_calls(few, s, 2000, drain), time_calls(many, s, 2000, drain));
    if (fn) std::printf(" | hipModuleLaunchKernel, packed buffer %.2f us", time_calls(module, s, 2000, drain));
    std::printf("\n");
  }
  {
    double tot = 0;
    for (int i = 0; i < 2000; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      (void)hipGetLastError();
      tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::printf("hipGetLastError: %.3f us\n", tot / 2000);
    // the launch after 60 us of spinning on a host word (what the iteration loop does between launches)
    volatile unsigned long long word = 0;
    double t2 = 0;
    for (int i = 0; i < 500; ++i) {
      (void)hipStreamSynchronize(s);
      const auto w0 = std::chrono::steady_clock::now();
      while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() < 60.0) word = word + 1;
      const auto t0 = std::chrono::steady_clock::now();
      many();
      t2 += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::printf("17-arg launch after 60 us of spinning: %.2f us\n", t2 / 500);
  }
  return 0;
}
