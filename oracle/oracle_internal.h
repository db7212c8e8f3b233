/* oracle_internal.h -- shared between the oracle's translation units (test infrastructure only). */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include <stddef.h>
#include <stdint.h>
void* orc_kd_build(const float* pts_xyzw, size_t n, int arith);
void orc_kd_free(void* tree);
void orc_kd_nearest(const void* tree, const float* q, int32_t* idx, float* d2);
/* the k smallest (d2, index) keys in ascending order; returns how many were found (min(k, n)) */
int orc_kd_knn(const void* tree, const float* q, int k, int32_t* idx, float* d2);
void orc_svd3(const double A[9], double U[9], double s[3], double V[9]);
#endif
