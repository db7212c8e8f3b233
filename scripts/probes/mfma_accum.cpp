// mfma_accum.cpp -- how exactly does v_mfma_f32_32x32x16_bf16 add its products?  (icp_brute_bf16.hip's error
// budget ASSUMED 2^-17 (P^2 + |v|^2) for "the MFMA's own accumulation".)  Every bf16 x bf16 product is exact in float32 (8 x 8
// significant bits); what is not documented is the adder: width, alignment, where it rounds.  This probe feeds adversarial
// operands -- cancelling products of the largest magnitude in the 14 K slots the kernel uses, exponent spreads from 2^0 to 2^-30,
// random signs -- and compares every one of the 1024 outputs of each MFMA with the exact sum (float64 holds it: 16 terms of 16
// significant bits over < 40 binades).  Reported: the worst |result - exact| relative to sum |terms| (the quantity the bound's
// budget is written in) and relative to max |term|.      hipcc --offload-arch=gfx950 -O2 mfma_accum.cpp -o mfma_accum
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void run(const unsigned short* A, const unsigned short* B, float* C, int n_mfma) {  // A: [n][32][16], B: [n][16][32] (bf16 bits)
  const int l = threadIdx.x;
  for (int t = blockIdx.x; t < n_mfma; t += gridDim.x) {
    bf16x8 a, b;
    unsigned short ua[8], ub[8];
    for (int e = 0; e < 8; ++e) {
      ua[e] = A[((size_t)t * 32 + (l & 31)) * 16 + 8 * (l >> 5) + e];
      ub[e] = B[((size_t)t * 16 + 8 * (l >> 5) + e) * 32 + (l & 31)];
    }
    memcpy(&a, ua, 16);
    memcpy(&b, ub, 16);
    floatx16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r / 4) + 4 * (l >> 5) + (r % 4), j = l & 31;
      C[((size_t)t * 32 + i) * 32 + j] = c[r];
    }
  }
}
static unsigned short f2bf(float f) {  // round to nearest even
  unsigned int u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static double bf2d(unsigned short h) {
  unsigned int u = (unsigned int)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return (double)f;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<unsigned short> A((size_t)n * 32 * 16), B((size_t)n * 16 * 32);
  const char* names[] = {"random, one binade", "random, 30 binades of spread", "cancelling pairs of the largest terms + small rest", "14 slots used (kernel's layout), cancelling",
                         "one huge +/- pair, tiny rest (2^-24 .. 2^-36 of it)"};
  double worst_sum[5] = {0}, worst_max[5] = {0}, worst_res[5] = {0};
  for (int t = 0; t < n; ++t) {
    const int pat = t % 5;
    for (int i = 0; i < 32; ++i)
      for (int k = 0; k < 16; ++k) {
        double v = U(rng);
        if (pat == 1 || pat == 2 || pat == 3) v = std::ldexp(v, -(int)(rng() % 15));
        if ((pat == 3) && k >= 14) v = 0.0;
        if (pat == 4) v = k < 2 ? 1.0 + 0.5 * U(rng) : std::ldexp(v, -12 - (int)(rng() % 6));
        A[((size_t)t * 32 + i) * 16 + k] = f2bf((float)v);
      }
    for (int j = 0; j < 32; ++j)
      for (int k = 0; k < 16; ++k) {
        double v = U(rng);
        if (pat == 1) v = std::ldexp(v, -(int)(rng() % 15));
        if (pat == 2 || pat == 3) {  // slot 2m+1 cancels slot 2m as far as the operands allow: b' = -b a / a' needs a' -- instead use equal a's below
          v = (k & 1) ? -std::fabs(v) : std::fabs(v);
        }
        if (pat == 3 && k >= 14) v = 0.0;
        if (pat == 4) v = k == 0 ? 1.5 : (k == 1 ? -1.5 : std::ldexp(v, -12 - (int)(rng() % 6)));
        B[((size_t)t * 16 + k) * 32 + j] = f2bf((float)v);
      }
    if (pat == 2 || pat == 3 || pat == 4)  // make the A operands of a cancelling slot pair EQUAL, so products cancel to the operands' last bits
      for (int i = 0; i < 32; ++i)
        for (int k = 0; k + 1 < (pat == 4 ? 2 : 16); k += 2) A[((size_t)t * 32 + i) * 16 + k + 1] = A[((size_t)t * 32 + i) * 16 + k];
    if (pat == 2 || pat == 3)
      for (int j = 0; j < 32; ++j)
        for (int k = 0; k + 1 < 16; k += 2) {  // |b_{k+1}| = |b_k| (1 + small): a residue far below the terms
          const double bk = bf2d(B[((size_t)t * 16 + k) * 32 + j]);
          B[((size_t)t * 16 + k + 1) * 32 + j] = f2bf((float)(-bk * (1.0 + ((rng() % 3) ? 0.0 : 0.0078125))));
        }
  }
  unsigned short *dA, *dB; float* dC;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, (size_t)n * 1024 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run, dim3(256), dim3(64), 0, 0, dA, dB, dC, n);
  std::vector<float> C((size_t)n * 1024);
  hipError_t e = hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; }
  double map_err = 0;
  for (int t = 0; t < n; ++t) {
    const int pat = t % 5;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double exact = 0, sum_abs = 0, mx = 0;
        for (int k = 0; k < 16; ++k) {
          const double p = bf2d(A[((size_t)t * 32 + i) * 16 + k]) * bf2d(B[((size_t)t * 16 + k) * 32 + j]);
          exact += p; sum_abs += std::fabs(p); mx = std::max(mx, std::fabs(p));
        }
        const double got = (double)C[((size_t)t * 32 + i) * 32 + j];
        const double err = std::fabs(got - exact);
        if (sum_abs > 0) {
          worst_sum[pat] = std::max(worst_sum[pat], err / sum_abs);
          worst_max[pat] = std::max(worst_max[pat], err / mx);
          if (exact != 0) worst_res[pat] = std::max(worst_res[pat], err / std::fabs(exact));
          if (pat == 0) map_err = std::max(map_err, err / sum_abs);
        }
      }
  }
  printf("%d MFMAs of v_mfma_f32_32x32x16_bf16, %d outputs each compared with the exact sum (operand layout check: %.1e)\n", n, 1024, map_err);
  for (int p = 0; p < 5; ++p)
    printf("  %-62s worst |err| / sum|terms| = %.3e = 2^%.1f ; / max|term| = 2^%.1f ; / |exact result| = %.2e\n", names[p], worst_sum[p], std::log2(worst_sum[p] > 0 ? worst_sum[p] : 1e-300),
           std::log2(worst_max[p] > 0 ? worst_max[p] : 1e-300), worst_res[p]);
  return 0;
}
