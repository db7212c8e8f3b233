// Dev probe (round 3): issue rate of single VALU instruction classes on gfx950, all CUs busy, 8 waves per SIMD.
// VERDICT round 2 corrected the issue roofline to SIMD-32 / 2 cycles per wave64 instruction (the guide's v_fma_f32 figure);
// nn_quad_kernel's mix is integer compares, selects, DPP moves and lane reads, not FMAs -- this measures what THOSE cost.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/probes/valu_rate.cpp -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kUnroll = 64;   // instructions per loop trip (8 independent chains of 8)
constexpr int kTrips = 2000;

// one instruction per chain step; r0..r7 are independent accumulators so that the dependent-issue latency does not bind
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL(NAME, OP)                                                                                     \
  __global__ __launch_bounds__(256) void NAME(float* out, int trips) {                                       \
    float r[8];                                                                                              \
    for (int k = 0; k < 8; ++k) r[k] = (float)(threadIdx.x + k);                                             \
    float a = out[0], b = out[1];                                                                            \
    unsigned int sel = threadIdx.x & 1;                                                                      \
    (void)sel;                                                                                               \
    for (int t = 0; t < trips; ++t) { BODY(OP) }                                                             \
    float s = 0.f;                                                                                           \
    for (int k = 0; k < 8; ++k) s += r[k];                                                                   \
    if (s == 123.456f) out[threadIdx.x] = s;                                                                 \
  }

#define OP_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_ADDF(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_SUBF(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_MULF(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_MINU(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(r[i]) : "v"(a));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : );
#define OP_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define OP_CMP64(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(*(unsigned long long*)&r[i & 6]), "v"(*(unsigned long long*)&r[(i + 2) & 6]) : "vcc");
#define OP_CMPE64(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : : "v"(r[i]), "v"(a) : "s20", "s21");
#define OP_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define OP_MINDPP(i) asm volatile("v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define OP_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(r[i]) : "s20");
#define OP_MIN3(i) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(double*)&r[i & 6]) : "v"(*(double*)&r[(i + 2) & 6]));
#define OP_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
#define OP_CVT(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
#define OP_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_SALU(i) asm volatile("s_add_u32 s20, s20, 1" : : : "s20");
#define OP_F64(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(*(double*)&r[i & 6]) : "v"(*(double*)&r[(i + 2) & 6]));

#define OP_CND_E64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(a) : );
#define OP_CND_AB(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(a), "v"(b) : );
#define OP_CND_CONST(i) asm volatile("v_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(r[i]) : : );
#define OP_CMP_CND(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_MAXU(i) asm volatile("v_max_u32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_ASHR(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(r[i]));
#define OP_SUBU(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(4)" : "+v"(r[i]) : "v"(a));
#define OP_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));
#define OP_MAXF(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_LSHL(i) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(r[i]));
#define OP_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_SUBCO(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc");
#define OP_PAIR_E64(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(r[i]) : "v"(a), "v"(b) : "s20", "s21");
#define OP_PAIR_E32(i) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define OP_CND_E64_VCC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : );
#define OP_CND_SMOV(i) asm volatile("s_mov_b64 vcc, exec\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
#define OP_ONE_THREE(i) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %2, %0, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(a), "v"(b) : "s20", "s21");
#define OP_MASKSEL(i) asm volatile("v_sub_u32 %0, %0, %1\n\tv_ashrrev_i32 %0, 31, %0\n\tv_and_b32 %0, %0, %2\n\tv_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a), "v"(b));
KERNEL(k_pair_e64, OP_PAIR_E64)
KERNEL(k_pair_e32, OP_PAIR_E32)
KERNEL(k_cnd_e64_vcc, OP_CND_E64_VCC)
KERNEL(k_cnd_smov, OP_CND_SMOV)
KERNEL(k_one_three, OP_ONE_THREE)
KERNEL(k_masksel, OP_MASKSEL)
KERNEL(k_cnd_e64, OP_CND_E64)
KERNEL(k_cnd_ab, OP_CND_AB)
KERNEL(k_cnd_const, OP_CND_CONST)
KERNEL(k_cmp_cnd, OP_CMP_CND)
KERNEL(k_bfi, OP_BFI)
KERNEL(k_and, OP_AND)
KERNEL(k_maxu, OP_MAXU)
KERNEL(k_ashr, OP_ASHR)
KERNEL(k_subu, OP_SUBU)
KERNEL(k_med3, OP_MED3)
KERNEL(k_mullo, OP_MULLO)
KERNEL(k_mad24, OP_MAD24)
KERNEL(k_bperm, OP_BPERM)
KERNEL(k_floor, OP_FLOOR)
KERNEL(k_maxf, OP_MAXF)
KERNEL(k_cvti, OP_CVTI)
KERNEL(k_add3, OP_ADD3)
KERNEL(k_lshl, OP_LSHL)
KERNEL(k_or3, OP_OR3)
KERNEL(k_subco, OP_SUBCO)
KERNEL(k_fma, OP_FMA)
KERNEL(k_fmac, OP_FMAC)
KERNEL(k_addf, OP_ADDF)
KERNEL(k_subf, OP_SUBF)
KERNEL(k_mulf, OP_MULF)
KERNEL(k_addu, OP_ADDU)
KERNEL(k_minu, OP_MINU)
KERNEL(k_lshladd, OP_LSHLADD)
KERNEL(k_mov, OP_MOV)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_cmp, OP_CMP)
KERNEL(k_cmp64, OP_CMP64)
KERNEL(k_cmpe64, OP_CMPE64)
KERNEL(k_dpp, OP_DPP)
KERNEL(k_mindpp, OP_MINDPP)
KERNEL(k_readlane, OP_READLANE)
KERNEL(k_min3, OP_MIN3)
KERNEL(k_pkfma, OP_PKFMA)
KERNEL(k_sqrt, OP_SQRT)
KERNEL(k_cvt, OP_CVT)
KERNEL(k_salu, OP_SALU)
KERNEL(k_f64, OP_F64)

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* d = nullptr;
  CK(hipMalloc(&d, 4096));
  CK(hipMemset(d, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct { const char* name; void (*fn)(float*, int); } tests[] = {
      {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_add_f32", k_addf}, {"v_sub_f32", k_subf}, {"v_mul_f32", k_mulf},
      {"v_min3_f32", k_min3}, {"v_pk_fma_f32", k_pkfma}, {"v_fma_f64", k_f64}, {"v_add_u32", k_addu}, {"v_min_u32", k_minu},
      {"v_lshl_add_u32", k_lshladd}, {"v_mov_b32", k_mov}, {"v_cndmask_b32 (vcc)", k_cndmask}, {"v_cmp_lt_u32 (vcc)", k_cmp},
      {"v_cmp_lt_u64 (vcc)", k_cmp64}, {"v_cmp_lt_u32_e64 (sgpr pair)", k_cmpe64}, {"v_mov_b32_dpp quad_perm", k_dpp},
      {"v_min_u32_dpp row_mirror", k_mindpp}, {"v_readlane_b32", k_readlane}, {"v_sqrt_f32", k_sqrt}, {"v_cvt_f32_i32", k_cvt},
      {"s_add_u32", k_salu}, {"v_cndmask_b32_e64 (sgpr pair)", k_cnd_e64}, {"v_cndmask_b32 d,a,b (vcc)", k_cnd_ab},
      {"v_cndmask_b32_e64 d,0,1 (vcc)", k_cnd_const}, {"v_cmp_lt_u32 + v_cndmask (pair = 2)", k_cmp_cnd}, {"v_bfi_b32", k_bfi},
      {"v_and_b32", k_and}, {"v_max_u32", k_maxu}, {"v_ashrrev_i32", k_ashr}, {"v_sub_u32", k_subu}, {"v_med3_i32", k_med3},
      {"v_mul_lo_u32", k_mullo}, {"v_mad_u32_u24", k_mad24}, {"ds_bpermute_b32", k_bperm}, {"v_floor_f32", k_floor},
      {"v_max_f32", k_maxf}, {"v_cvt_i32_f32", k_cvti}, {"v_add3_u32", k_add3}, {"v_lshlrev_b32", k_lshl}, {"v_or3_b32", k_or3},
      {"v_sub_co_u32 (vcc out)", k_subco}, {"PAIR cmp_e64 s[20:21] + nop + cnd_e64 (x2)", k_pair_e64},
      {"PAIR cmp_e32 vcc + nop + cnd_e32 (x2)", k_pair_e32}, {"v_cndmask_b32_e64 d,d,a,vcc", k_cnd_e64_vcc},
      {"s_mov vcc + v_cndmask_e32 (x1 valu)", k_cnd_smov}, {"cmp_e64 + 3 cnd_e64 (x4)", k_one_three},
      {"sub+ashr+and+add select (x4)", k_masksel}};
  printf("CUs %d, clock %d kHz; grid = CUs x 8 workgroups of 256 (8 waves per SIMD), %d x %d instructions per wave\n", cus,
         prop.clockRate, kTrips, kUnroll);
  for (int waves_per_simd : {8, 1}) {
    printf("-- %d wave(s) per SIMD\n", waves_per_simd);
    for (auto& t : tests) {
      const int blocks = cus * waves_per_simd;  // 4 waves per workgroup = one per SIMD
      hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, d, 10);
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, d, kTrips);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double insts = (double)blocks * 4.0 * kTrips * kUnroll;  // wave-level instructions
      const double ginst = insts / (best * 1e-3) / 1e9;
      const double cyc = (double)cus * 4.0 * 2.4e9 * (best * 1e-3) / insts;  // SIMD-cycles per instruction at 2.4 GHz
      printf("%-32s %8.3f ms  %8.1f G wave-instr/s  %5.2f cycles per instruction and SIMD (at 2.4 GHz)\n", t.name, best, ginst, cyc);
    }
  }
  return 0;
}
