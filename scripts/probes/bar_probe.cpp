// Dev probe: can the CPU write into fine-grained DEVICE memory (through the PCIe BAR), and how fast does a resident kernel
// see it?  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/probes/bar_probe.cpp -o /tmp/bar_probe && /tmp/bar_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

// echo server: waits for cmd[0] == expect, writes it to the host mailbox
__global__ void echo_kernel(volatile unsigned int* cmd, volatile unsigned int* host_out, unsigned int first, unsigned int n) {
  unsigned int expect = first;
  const long long t0 = (long long)wall_clock64();
  for (unsigned int k = 0; k < n; ++k) {
    for (;;) {
      const unsigned int v = __hip_atomic_load((unsigned int*)cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v == expect) break;
      if ((long long)wall_clock64() - t0 > 200000000ll) return;  // 2 s
    }
    __hip_atomic_store((unsigned int*)host_out, expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ++expect;
  }
}

static int round_trips(const char* name, volatile unsigned int* cmd_host_view, unsigned int* cmd_dev_view, volatile unsigned int* out_host,
                       unsigned int* out_dev, hipStream_t st) {
  const unsigned int n = 2000, first = 1000;
  *out_host = 0;
  hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(1), 0, st, cmd_dev_view, out_dev, first, n);
  auto t0 = std::chrono::steady_clock::now();
  for (unsigned int k = 0; k < n; ++k) {
    *cmd_host_view = first + k;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    unsigned long spins = 0;
    while (*out_host != first + k)
      if (++spins > 400000000ul) { printf("%s: no answer\n", name); return 1; }
  }
  auto t1 = std::chrono::steady_clock::now();
  CK(hipStreamSynchronize(st));
  printf("%s: %.2f us per host->device->host round trip\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
  return 0;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned int *out_host = nullptr, *out_dev = nullptr;
  CK(hipHostMalloc((void**)&out_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&out_dev, out_host, 0));
  // (a) command line in mapped host memory (what gicp_server_kernel does today)
  unsigned int *cmd_host = nullptr, *cmd_host_dev = nullptr;
  CK(hipHostMalloc((void**)&cmd_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&cmd_host_dev, cmd_host, 0));
  *cmd_host = 0;
  if (round_trips("command in host memory  ", cmd_host, cmd_host_dev, out_host, out_dev, st)) return 1;
  // (b) command line in fine-grained device memory, written by the CPU through the BAR
  unsigned int* cmd_vram = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&cmd_vram, 4096, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  CK(hipMemset(cmd_vram, 0, 4096));
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  if (sigsetjmp(jb, 1)) { printf("CPU access to fine-grained device memory: FAULT\n"); return 0; }
  volatile unsigned int* v = cmd_vram;
  *v = 7;
  printf("CPU wrote fine-grained device memory, reads back %u\n", *v);
  unsigned int chk = 0;
  CK(hipMemcpy(&chk, cmd_vram, 4, hipMemcpyDeviceToHost));
  printf("device copy sees %u\n", chk);
  *v = 0;
  if (round_trips("command in device memory", v, cmd_vram, out_host, out_dev, st)) return 1;
  return 0;
}
