// trig_check.cpp -- icpslam_amd/csrc/icp_trig.h (the product's correctly rounded sin / cos, double-double arithmetic) against binary128
// (libquadmath, what the oracle's EXACT mode uses): counts the arguments on which the two differ.  tests/test_trig.py builds and runs it.
#include <quadmath.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "icp_trig.h"
int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 400000;
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
  long bs = 0, bc = 0, bsf = 0, bcf = 0, libm_s = 0, libm_sf = 0, far = 0;
  for (long i = 0; i < n; ++i) {
    const double scale = (i % 4 == 0) ? 0.05 : (i % 4 == 1) ? 0.5 : (i % 4 == 2) ? 3.3 : 200.0;
    double x = (rnd() * 2 - 1) * scale;
    if (i % 97 == 0) x = (double)(long)(x * 81.48733086305041) * 0.012271846303085129797 + (rnd() - 0.5) * 1e-9;  // near multiples of pi / 256
    double s, c;
    icpgpu::trig::sincos_cr(x, &s, &c);
    const double sq = (double)sinq((__float128)x), cq = (double)cosq((__float128)x);
    bs += s != sq; bc += c != cq;
    if (std::fabs(s - sq) > 2.3e-16 * std::fabs(sq) + 1e-300 || std::fabs(c - cq) > 2.3e-16 * std::fabs(cq) + 1e-300) ++far;  // never more than an ulp
    libm_s += std::sin(x) != sq;
    const float xf = (float)x;
    float sf, cf;
    icpgpu::trig::sincosf_cr(xf, &sf, &cf);
    bsf += sf != (float)sinq((__float128)xf); bcf += cf != (float)cosq((__float128)xf);
    libm_sf += sinf(xf) != (float)sinq((__float128)xf);
  }
  printf("%ld %ld %ld %ld %ld %ld %ld %ld\n", n, bs, bc, bsf, bcf, far, libm_s, libm_sf);
  return 0;
}
