// Dev probe (round 5): a 3.2 MB cloud from PAGEABLE host memory into HBM -- what icpgpu_set_source does per scan.
//   (a) hipMemcpyAsync(pageable -> device) + hipStreamSynchronize   (the runtime's own staging)
//   (b) T host threads copy slices into a pinned staging buffer, each slice goes out with hipMemcpyAsync as soon as it is copied
// Build: hipcc --offload-arch=gfx950 -O2 -pthread -o upload_probe upload_probe.cpp
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : 3200000;
  const int reps = 200;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  char* dev; CK(hipMalloc(&dev, bytes));
  char* pinned; CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
  // several pageable sources, rotated (a fresh scan is not necessarily in the cache)
  const int n_src = 6;
  std::vector<char*> src(n_src);
  for (auto& p : src) { p = (char*)malloc(bytes); memset(p, 1, bytes); }
  std::vector<double> t;
  auto report = [&](const char* name) {
    std::sort(t.begin(), t.end());
    printf("%-64s median %7.1f us  p10 %7.1f  p90 %7.1f   (%.1f GB/s)\n", name, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], bytes / t[t.size() / 2] * 1e-3);
    t.clear();
  };
  for (int r = 0; r < reps; ++r) {
    auto a = clk::now();
    CK(hipMemcpyAsync(dev, src[r % n_src], bytes, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    t.push_back(us(a, clk::now()));
  }
  report("(a) hipMemcpyAsync pageable + synchronize");
  for (int r = 0; r < reps; ++r) {
    auto a = clk::now();
    CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    t.push_back(us(a, clk::now()));
  }
  report("(ref) hipMemcpyAsync from PINNED + synchronize (the DMA alone)");
  for (int r = 0; r < reps; ++r) {
    auto a = clk::now();
    memcpy(pinned, src[r % n_src], bytes);
    t.push_back(us(a, clk::now()));
  }
  report("(ref) one thread's memcpy pageable -> pinned");
  // persistent helpers, spinning (the best case for the hand-over; a sleeping helper adds its wake-up)
  for (int T : {1, 2, 3, 4}) for (int slices_per_thread : {1, 2, 4}) {
    const int S = T * slices_per_thread;
    std::atomic<int> go{0}, next{0}, done{0};
    std::atomic<bool> quit{false};
    const char* cur = nullptr;
    auto body = [&]() {
      for (;;) {
        const int s = next.fetch_add(1);
        if (s >= S) break;
        const size_t lo = bytes * s / S / 64 * 64, hi = s + 1 == S ? bytes : bytes * (s + 1) / S / 64 * 64;
        memcpy(pinned + lo, cur + lo, hi - lo);
        (void)hipMemcpyAsync(dev + lo, pinned + lo, hi - lo, hipMemcpyHostToDevice, st);
        done.fetch_add(1);
      }
    };
    std::vector<std::thread> helpers;
    for (int k = 1; k < T; ++k)
      helpers.emplace_back([&] {
        int seen = 0;
        while (!quit.load(std::memory_order_relaxed)) {
          if (go.load(std::memory_order_acquire) != seen) { seen++; body(); }
          else __builtin_ia32_pause();
        }
      });
    for (int r = 0; r < reps; ++r) {
      auto a = clk::now();
      cur = src[r % n_src];
      next.store(0); done.store(0);
      go.fetch_add(1, std::memory_order_release);
      body();
      while (done.load() < S) __builtin_ia32_pause();
      CK(hipStreamSynchronize(st));
      t.push_back(us(a, clk::now()));
    }
    quit.store(true);
    for (auto& h : helpers) h.join();
    char name[128];
    snprintf(name, sizeof name, "(b) %d thread(s), %d slices: memcpy -> pinned, DMA per slice", T, S);
    report(name);
  }
  return 0;
}
