// solve_probe.cpp -- gicp_solve_kernel on a synthetic problem, with the slots inspected from the host afterwards.
//   hipcc --offload-arch=gfx950 -O2 -I icpslam_amd/csrc scripts/probes/solve_probe.cpp -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "icp_kernels.h"
using namespace icpgpu;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = (n + 1023) / 1024;
  std::mt19937 rng(5);
  std::normal_distribution<float> N01(0.f, 1.f);
  std::vector<float4> src(n), tgt(n);
  std::vector<unsigned long long> keys(n);
  std::vector<double> maha((size_t)n * 6);
  for (int i = 0; i < n; ++i) {
    src[i] = make_float4(10 * N01(rng), 10 * N01(rng), 2 * N01(rng), 1.f);
    tgt[i] = make_float4(src[i].x + 0.05f + 0.01f * N01(rng), src[i].y - 0.03f + 0.01f * N01(rng), src[i].z + 0.01f * N01(rng), 1.f);
    const float d2 = 0.01f;
    unsigned int db;
    memcpy(&db, &d2, 4);
    keys[i] = ((unsigned long long)db << 32) | (unsigned)i;
    double* M = &maha[(size_t)i * 6];
    M[0] = 1 + 0.1 * N01(rng); M[1] = 0.01; M[2] = 0.02; M[3] = 1.2; M[4] = 0.03; M[5] = 50;
  }
  float4 *d_src, *d_tgt; unsigned long long* d_keys; double* d_maha; unsigned long long *slots, *h_out, *h_out_dev;
  hipMalloc(&d_src, n * 16); hipMalloc(&d_tgt, n * 16); hipMalloc(&d_keys, n * 8); hipMalloc(&d_maha, (size_t)n * 48);
  hipMemcpy(d_src, src.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(d_tgt, tgt.data(), n * 16, hipMemcpyHostToDevice);
  hipMemcpy(d_keys, keys.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d_maha, maha.data(), (size_t)n * 48, hipMemcpyHostToDevice);
  const size_t sb = gicp_solve_slot_bytes(256);
  hipExtMallocWithFlags((void**)&slots, sb, hipDeviceMallocFinegrained); hipMemset(slots, 0, sb);
  hipHostMalloc((void**)&h_out, 512, hipHostMallocMapped); memset(h_out, 0, 512);
  hipHostGetDevicePointer((void**)&h_out_dev, h_out, 0);
  Xform base{}; base.m[0] = base.m[5] = base.m[10] = 1.f;
  float guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double x0[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long *local_slots, *owner;
  hipMalloc((void**)&local_slots, sb); hipMemset(local_slots, 0, sb);
  hipMalloc((void**)&owner, 256 * 8); hipMemset(owner, 0, 256 * 8);
  const int xcc = argc > 2 ? atoi(argv[2]) : 0;
  for (int rep = 0; rep < 6; ++rep) {
    const bool local = rep >= 3;
    const unsigned long long seq0 = 8192ull * (rep + 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipError_t e = launch_gicp_solve(blocks, d_src, n, d_tgt, d_keys, 1.0f, base, guess, d_maha, x0, slots, h_out_dev, seq0, 20, 1e-2, 0, local ? local_slots : nullptr, local ? owner : nullptr, xcc);
    hipEventRecord(e1, 0);
    hipError_t e2 = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    double out[24] = {0};
    int valid = 0;
    for (int k = 0; k < gicp_solve_out_granules(); ++k) valid += gicp_granule_read(h_out + 2 * k, seq0, &out[k]) ? 1 : 0;
    printf("%s rep %d: launch %s sync %s, %.3f ms, %d/%d result granules valid; status %.0f x = %.6f %.6f %.6f %.6f %.6f %.6f m %.0f evals %.0f dbg %.0f -> %.2f us per evaluation\n", local ? "one-XCD" : "any-XCD", rep,
           hipGetErrorString(e), hipGetErrorString(e2), ms, valid, gicp_solve_out_granules(), out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], out[10], out[11], out[10] > 0 ? ms * 1e3 / out[10] : 0.0);
    if (out[0] != 3) {
      const double ev = out[10] > 0 ? out[10] : 1;
      printf("   per evaluation (us): apply_state %.2f | accumulate %.2f | publish %.2f | gather %.2f | gradient %.2f | kernel total %.2f -> solver between evaluations %.2f; gather passes per evaluation %.2f\n", out[12] / ev,
             out[13] / ev, out[14] / ev, out[15] / ev, out[16] / ev, out[17] / ev, (out[17] - out[12] - out[13] - out[14] - out[15] - out[16]) / ev, out[18] / ev);
    }
    if (out[0] == 3) {  // inspect the slots
      std::vector<unsigned long long> s(sb / 8);
      hipMemcpy(s.data(), slots, sb, hipMemcpyDeviceToHost);
      int shown = 0;
      for (int par = 0; par < 2; ++par)
        for (int b = 0; b < blocks; ++b)
          for (int g = 0; g < 28; ++g) {
            const unsigned long long bits = s[((size_t)par * blocks + b) * 56 + 2 * g], tag = s[((size_t)par * blocks + b) * 56 + 2 * g + 1];
            const unsigned long long fold = (bits ^ (bits >> 24) ^ (bits >> 48)) & 0xFFFFFF;
            if ((tag & 0xFFFFFF) != fold && shown++ < 40) printf("  parity %d block %d granule %d: number %llu, checksum %llu vs %llu of the bits %016llx\n", par, b, g, tag >> 24, tag & 0xFFFFFF, fold, bits);
          }
      printf("  inconsistent granules: %d\n", shown);
    }
  }
  return 0;
}
