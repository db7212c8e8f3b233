/*
 * gicp_oracle.c -- CPU ORACLE for the GICP mode.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (see icp_oracle.h).  Placeholder until the GICP row is built.
 */
#include "icp_oracle.h"

int orc_gicp_align(const float* src, size_t n_s, const float* tgt, size_t n_t, const orc_params* P,
                   const float* guess, float* out_xyzw, int want_fitness, orc_result* res, orc_iter_trace* trace) {
  (void)src; (void)n_s; (void)tgt; (void)n_t; (void)P; (void)guess; (void)out_xyzw; (void)want_fitness; (void)res; (void)trace;
  return -2; /* not implemented yet */
}
