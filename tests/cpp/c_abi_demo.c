/* c_abi_demo.c -- the plain-C binding of INTEGRATION.md section 3, compiled as C99: what any FFI (cgo, JNI, ctypes) binds.
 * usage: c_abi_demo <src.bin> <n_src> <tgt.bin> <n_tgt>      (clouds: raw float32 x,y,z,pad records)
 * prints: converged iterations n_corr fitness T[16] (column-major), then the same again after promote + re-align */
#include <stdio.h>
#include <stdlib.h>

#include "icpgpu.h"

static float* load(const char* path, size_t n) {
  float* p = (float*)malloc((n ? n : 1) * 4 * sizeof(float));
  FILE* f = fopen(path, "rb");
  if (!f || !p) { perror(path); exit(2); }
  if (n && fread(p, 4 * sizeof(float), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  fclose(f);
  return p;
}

static void show(const icpgpu_result* r) {
  int i;
  printf("%d %d %u %.17g", r->converged, r->iterations, r->n_correspondences, r->fitness);
  for (i = 0; i < 16; ++i) printf(" %.9g", r->T[i]);
  printf("\n");
}

int main(int argc, char** argv) {
  icpgpu_ctx* ctx = NULL;
  icpgpu_params p;
  icpgpu_result r;
  size_t ns, nt;
  float *src, *tgt;
  if (argc < 5) return 2;
  ns = strtoull(argv[2], NULL, 10);
  nt = strtoull(argv[4], NULL, 10);
  src = load(argv[1], ns);
  tgt = load(argv[3], nt);
  if (icpgpu_create(&ctx, 0) != ICPGPU_OK) { fprintf(stderr, "%s\n", icpgpu_last_error(NULL)); return 3; }
  icpgpu_default_params(&p);            /* 10 iterations, epsilon 1e-6, gate 1.0 m: icp_odometer.h:63-65 */
  if (icpgpu_set_params(ctx, &p) != ICPGPU_OK) return 4;
  if (icpgpu_set_source(ctx, src, ns) != ICPGPU_OK || icpgpu_set_target(ctx, tgt, nt) != ICPGPU_OK) return 4;
  if (icpgpu_align(ctx, NULL, NULL, 1, &r) != ICPGPU_OK) { fprintf(stderr, "%s\n", icpgpu_last_error(ctx)); return 5; }
  show(&r);
  /* *prev_cloud_ = *curr_cloud_ (icp_odometer.cpp:209) without a copy, then the old target comes back as the new scan */
  if (icpgpu_promote_source_to_target(ctx) != ICPGPU_OK) return 6;
  if (icpgpu_set_source(ctx, tgt, nt) != ICPGPU_OK) return 6;
  if (icpgpu_align(ctx, NULL, NULL, 1, &r) != ICPGPU_OK) return 7;
  show(&r);
  icpgpu_destroy(ctx);
  free(src);
  free(tgt);
  return 0;
}
