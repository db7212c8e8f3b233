// Dev probe (round 3): do f32 MFMAs and ordinary vector instructions of the same SIMD execute side by side on gfx950?
// nn_brute_mfma_kernel's counters (profiles/r02_brute_force_mfma.txt) show matrix time (64 %) and vector time (27 %) ADDING UP,
// whether the folds run behind the wave's own MFMAs or under the next step's (profiles/r03_brute_force_pipelined.txt).  This
// measures the rule directly: a loop of 4 independent v_mfma_f32_32x32x2_f32 plus V independent vector instructions per trip,
// against each part alone, in one wave and spread over W waves per SIMD -- and the same with a bf16 MFMA for comparison.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_coissue.cpp -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef short shortx8 __attribute__((ext_vector_type(8)));

// MK: 0 none, 1 four v_mfma_f32_32x32x2_f32, 2 four v_mfma_f32_32x32x16_bf16;  VK: 0 v_fma_f32 (full rate), 1 v_min3_f32 (half rate)
template <int MK, int V, int VK>
__global__ __launch_bounds__(256) void probe(float* out, int trips) {
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  float r[8];
  for (int k = 0; k < 8; ++k) r[k] = (float)(threadIdx.x + k);
  const float a = out[0], b = out[1];
  shortx8 ha, hb;
  for (int k = 0; k < 8; ++k) { ha[k] = (short)threadIdx.x; hb[k] = (short)k; }
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MK == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MK == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int v = 0; v < V / 4; ++v) {
        if (VK == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[v & 7]) : "v"(a), "v"(b));
        else asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(r[v & 7]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += r[k];
  for (int i = 0; i < 4; ++i) s += acc[i][3];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// waves 0..3 of a 512-thread workgroup (one per SIMD) run the MFMAs, waves 4..7 (again one per SIMD) the vector instructions
template <int V, int VK>
__global__ __launch_bounds__(512) void probe_split(float* out, int trips, int roles) {
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  float r[8];
  for (int k = 0; k < 8; ++k) r[k] = (float)(threadIdx.x + k);
  const float a = out[0], b = out[1];
  const bool matrix_wave = threadIdx.x < 256;
  if (matrix_wave) {
    if (roles & 1)
      for (int t = 0; t < trips; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  } else if (roles & 2) {
    for (int t = 0; t < trips; ++t)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (VK == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[v & 7]) : "v"(a), "v"(b));
        else asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(r[v & 7]) : "v"(a), "v"(b));
      }
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += r[k];
  for (int i = 0; i < 4; ++i) s += acc[i][3];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// the search kernel's step in miniature: two bf16 MFMAs from zero, then 16 v_min3_f32 folding THEIR results (DEP = 1) or
// other registers (DEP = 0), optionally one step behind (PIPE = 1: this step's folds read the other accumulator pair)
template <int DEP, int PIPE>
__global__ __launch_bounds__(256) void probe_step(float* out, int trips) {
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 16; ++k) acc[i][k] = (float)(threadIdx.x + k);
  floatx16 other[2];
  for (int i = 0; i < 2; ++i)
    for (int k = 0; k < 16; ++k) other[i][k] = (float)(threadIdx.x * 3 + k);
  shortx8 ha, hb;
  for (int k = 0; k < 8; ++k) { ha[k] = (short)threadIdx.x; hb[k] = (short)k; }
  float m0 = 0.f, m1 = 0.f;
#define FOLD(A, B)                                                                                                            \
  asm volatile("v_min3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %5, %6, %7\n\tv_min3_f32 %0, %0, %8, %9\n\tv_min3_f32 %1, %1, %10, %11\n\t"   \
               "v_min3_f32 %0, %0, %12, %13\n\tv_min3_f32 %1, %1, %14, %15\n\tv_min3_f32 %0, %0, %16, %17\n\tv_min3_f32 %0, %0, %1, %0" \
               : "+v"(m0), "+v"(m1)                                                                                            \
               : "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]), "v"(A[8]), "v"(A[9]),   \
                 "v"(A[10]), "v"(A[11]), "v"(A[12]), "v"(A[13]), "v"(A[14]), "v"(A[15]));                                       \
  asm volatile("v_min3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %5, %6, %7\n\tv_min3_f32 %0, %0, %8, %9\n\tv_min3_f32 %1, %1, %10, %11\n\t"   \
               "v_min3_f32 %0, %0, %12, %13\n\tv_min3_f32 %1, %1, %14, %15\n\tv_min3_f32 %0, %0, %16, %17\n\tv_min3_f32 %0, %0, %1, %0" \
               : "+v"(m0), "+v"(m1)                                                                                            \
               : "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(B[6]), "v"(B[7]), "v"(B[8]), "v"(B[9]),   \
                 "v"(B[10]), "v"(B[11]), "v"(B[12]), "v"(B[13]), "v"(B[14]), "v"(B[15]));
  for (int t = 0; t < trips; ++t) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, 0" : "=v"(acc[0]), "=v"(acc[1]) : "v"(ha), "v"(hb));
    if (PIPE) {
      FOLD(acc[2], acc[3])
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %3, 0" : "=v"(acc[2]), "=v"(acc[3]) : "v"(ha), "v"(hb));
      asm volatile("s_nop 10");
      FOLD(acc[0], acc[1])
    } else if (DEP) {
      asm volatile("s_nop 10");
      FOLD(acc[0], acc[1])
    } else {
      FOLD(other[0], other[1])
    }
  }
  float s = m0 + m1;
  for (int i = 0; i < 4; ++i) s += acc[i][3];
  if (s == 123.456f) out[threadIdx.x] = s;
}

static float* g_buf;
static int g_cus;

template <typename F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

constexpr int kTrips = 4000;

template <int MK, int V, int VK>
static void run(const char* what) {
  for (int w = 1; w <= 4; w *= 2) {   // waves per SIMD (a 256-thread workgroup = one wave on each SIMD of a CU)
    const double ms = time_ms([&] { hipLaunchKernelGGL((probe<MK, V, VK>), dim3(g_cus * w), dim3(256), 0, 0, g_buf, kTrips); });
    printf("%-58s waves/SIMD %d: %8.3f ms = %7.1f ns per trip and SIMD\n", what, w, ms, ms * 1e6 / (kTrips * (double)w));
  }
}

template <int V, int VK>
static void run_split(const char* what) {
  for (int roles = 1; roles <= 3; ++roles) {
    const double ms = time_ms([&] { hipLaunchKernelGGL((probe_split<V, VK>), dim3(g_cus), dim3(512), 0, 0, g_buf, kTrips, roles); });
    printf("%-58s %-22s %8.3f ms = %7.1f ns per trip\n", what, roles == 1 ? "matrix waves only" : roles == 2 ? "vector waves only" : "both, side by side", ms,
           ms * 1e6 / kTrips);
  }
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  g_cus = p.multiProcessorCount;
  printf("%s, %d CUs, %d MHz\n", p.gcnArchName, g_cus, p.clockRate / 1000);
  CK(hipMalloc(&g_buf, 1 << 20));
  CK(hipMemset(g_buf, 0, 1 << 20));
  printf("# one wave does both: 4 MFMAs + V vector instructions per trip (all independent)\n");
  run<1, 0, 0>("4 x v_mfma_f32_32x32x2_f32");
  run<0, 32, 0>("32 x v_fma_f32");
  run<0, 32, 1>("32 x v_min3_f32");
  run<1, 8, 0>("4 x mfma f32 + 8 x v_fma_f32");
  run<1, 16, 0>("4 x mfma f32 + 16 x v_fma_f32");
  run<1, 32, 0>("4 x mfma f32 + 32 x v_fma_f32");
  run<1, 64, 0>("4 x mfma f32 + 64 x v_fma_f32");
  run<1, 16, 1>("4 x mfma f32 + 16 x v_min3_f32");
  run<1, 32, 1>("4 x mfma f32 + 32 x v_min3_f32");
  run<2, 0, 0>("4 x v_mfma_f32_32x32x16_bf16");
  run<2, 16, 0>("4 x mfma bf16 + 16 x v_fma_f32");
  run<2, 32, 0>("4 x mfma bf16 + 32 x v_fma_f32");
  run<2, 32, 1>("4 x mfma bf16 + 32 x v_min3_f32");
  printf("# the search kernel's step: 2 bf16 MFMAs from zero + 16 v_min3_f32 per trip (PIPE: 4 + 32 per trip)\n");
  for (int w = 1; w <= 8; w *= 2) {
    const double a = time_ms([&] { hipLaunchKernelGGL((probe_step<0, 0>), dim3(g_cus * w), dim3(256), 0, 0, g_buf, kTrips); });
    const double b = time_ms([&] { hipLaunchKernelGGL((probe_step<1, 0>), dim3(g_cus * w), dim3(256), 0, 0, g_buf, kTrips); });
    const double c = time_ms([&] { hipLaunchKernelGGL((probe_step<1, 1>), dim3(g_cus * w), dim3(256), 0, 0, g_buf, kTrips); });
    printf("waves/SIMD %d: folds of other registers %6.1f ns per step and SIMD | folds of the MFMAs' results %6.1f | the same one step behind %6.1f\n", w,
           a * 1e6 / (kTrips * (double)w), b * 1e6 / (kTrips * (double)w), c * 1e6 / (2.0 * kTrips * (double)w));
  }
  printf("# separate waves of the same SIMD: one runs 4 MFMAs per trip, the other V vector instructions per trip\n");
  run_split<32, 0>("4 x mfma f32 | 32 x v_fma_f32");
  run_split<64, 0>("4 x mfma f32 | 64 x v_fma_f32");
  run_split<32, 1>("4 x mfma f32 | 32 x v_min3_f32");
  return 0;
}
