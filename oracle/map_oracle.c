/*
 * map_oracle.c -- CPU ORACLE for the mapper's map.  TEST INFRASTRUCTURE ONLY (see icp_oracle.h:
 * PARITY UNPINNED -- the octree is PCL's, which is not under /root/reference).
 *
 * Restates what the reference relies on at
 *   /root/reference/src/icpslam/octree_mapper.cpp:55-59   resetMap (OctreePointCloudSearch, resolution 0.5 m)
 *   /root/reference/src/icpslam/octree_mapper.cpp:62-69   addPointsToMap: "if (!isVoxelOccupiedAtPoint(p)) addPointToCloud(p)"
 *   /root/reference/src/icpslam/octree_mapper.cpp:72-90   approxNearestNeighbors -> nn cloud
 * as a plain sequential loop: points are visited in order; a point is appended iff its voxel is empty; a voxel is
 * floor((p - origin) / resolution) per axis in double, origin = the minimum corner of the octree's first bounding box = first
 * point ever added - resolution (PCL OctreePointCloud::adoptBoundingBoxToPoint sets the box to p +- resolution / 2 and calls
 * getKeyBitSize(), which -- max_voxels = max(ceil(extent / resolution), 2) -- makes the tree one level deep, 2 voxels wide, and
 * splits the oversize evenly: p +- resolution; genOctreeKeyforPoint indexes from that minimum; the box only grows by whole
 * octree side lengths afterwards, so the lattice never moves).  Until round 4 this file stopped at p - resolution / 2.  The nearest-neighbour query is EXACT (orc_nn), where PCL's approxNearestSearch is a
 * heuristic descent -- SURVEY.md 8(f4) asks for the exact one.
 *
 * The voxel set is a sorted array + binary search: deliberately nothing like the GPU's hash set.  Two forms of addPointsToMap:
 * orc_map_add_points_sequential is the reference's loop as written (one point at a time, O(map) per insertion: fine to ~100k
 * map points); orc_map_add_points states the same rule per batch -- a point is appended iff its voxel is not in the map before
 * the call AND no earlier point of the call has it -- with one sort of the batch's (voxel, index) pairs, so that a 1M-point map
 * (BASELINE config 3's scale) takes seconds.  tests/test_map_oracle.py holds the two against each other.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "icp_oracle.h"

struct orc_map {
  double res, ox, oy, oz;
  int anchored;
  size_t n, cap;
  float* pts;     /* n x 4 */
  int64_t* keys;  /* sorted voxel keys of the n points */
};

static int64_t pack_key(int64_t kx, int64_t ky, int64_t kz) { return ((kz + (1 << 20)) << 42) | ((ky + (1 << 20)) << 21) | (kx + (1 << 20)); }

static int point_key(const orc_map* m, const float* p, int64_t* key) {
  const double fx = floor(((double)p[0] - m->ox) / m->res), fy = floor(((double)p[1] - m->oy) / m->res),
               fz = floor(((double)p[2] - m->oz) / m->res);
  const double lim = (double)((1 << 20) - 1);
  if (!(fabs(fx) <= lim && fabs(fy) <= lim && fabs(fz) <= lim)) return 0;
  *key = pack_key((int64_t)fx, (int64_t)fy, (int64_t)fz);
  return 1;
}

/* adoptBoundingBoxToPoint on an empty octree followed by getKeyBitSize(), operation for operation (PCL 1.8
 * octree_pointcloud.hpp): returns the box's minimum corner, the lattice origin */
static void orc_octree_first_box(const float* p, double res, double* ox, double* oy, double* oz) {
  const float min_value = 1.1920928955078125e-7f; /* std::numeric_limits<float>::epsilon() */
  double mn[3], mx[3];
  unsigned int max_key = 0;
  for (int a = 0; a < 3; ++a) {
    mn[a] = (double)p[a] - res / 2;
    mx[a] = (double)p[a] + res / 2;
    const unsigned int k = (unsigned int)ceil((mx[a] - mn[a] - min_value) / res);
    if (k > max_key) max_key = k;
  }
  const unsigned int max_voxels = max_key > 2u ? max_key : 2u;
  const unsigned int depth = (unsigned int)ceil(log((double)max_voxels) / log(2.0) - min_value);
  const double side = (double)(1u << depth) * res;
  for (int a = 0; a < 3; ++a) {
    const double oversize = (side - (mx[a] - mn[a])) / 2.0;
    if (oversize > min_value) {
      mn[a] -= oversize;
      mx[a] += oversize;
    }
  }
  *ox = mn[0];
  *oy = mn[1];
  *oz = mn[2];
}

orc_map* orc_map_create(double resolution) {
  orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
  if (m) m->res = resolution;
  return m;
}

void orc_map_destroy(orc_map* m) {
  if (!m) return;
  free(m->pts);
  free(m->keys);
  free(m);
}

size_t orc_map_size(const orc_map* m) { return m->n; }
const float* orc_map_points(const orc_map* m) { return m->pts; }

static long find_key(const int64_t* keys, size_t n, int64_t k) { /* position of k in the sorted array or -(insert)-1 */
  size_t lo = 0, hi = n;
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (keys[mid] < k) lo = mid + 1;
    else hi = mid;
  }
  return (lo < n && keys[lo] == k) ? (long)lo : -(long)lo - 1;
}

/* addPointsToMap(transformCloudToPoseFrame(cloud, pose)); returns the number of points appended, -1 on allocation failure */
long orc_map_add_points_sequential(orc_map* m, const float* in_xyzw, size_t n, const float pose[16]) {
  float* moved = (float*)malloc((n ? n : 1) * 4 * sizeof(float));
  if (!moved) return -1;
  if (pose) orc_transform_cloud(in_xyzw, n, pose, moved);
  else {
    memcpy(moved, in_xyzw, n * 4 * sizeof(float));
    for (size_t i = 0; i < n; ++i) moved[4 * i + 3] = 1.0f;
  }
  long added = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* p = moved + 4 * i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    if (!m->anchored) {
      orc_octree_first_box(p, m->res, &m->ox, &m->oy, &m->oz);
      m->anchored = 1;
    }
    int64_t key;
    if (!point_key(m, p, &key)) continue;
    const long pos = find_key(m->keys, m->n, key);
    if (pos >= 0) continue; /* isVoxelOccupiedAtPoint */
    if (m->n == m->cap) {
      const size_t cap = m->cap ? 2 * m->cap : 4096;
      float* np_ = (float*)realloc(m->pts, cap * 4 * sizeof(float));
      int64_t* nk = (int64_t*)realloc(m->keys, cap * sizeof(int64_t));
      if (np_) m->pts = np_;
      if (nk) m->keys = nk;
      if (!np_ || !nk) {
        free(moved);
        return -1;
      }
      m->cap = cap;
    }
    const size_t ins = (size_t)(-pos - 1);
    memmove(m->keys + ins + 1, m->keys + ins, (m->n - ins) * sizeof(int64_t));
    m->keys[ins] = key;
    memcpy(m->pts + 4 * m->n, p, 4 * sizeof(float)); /* map_cloud_ keeps insertion order */
    m->n++;
    added++;
  }
  free(moved);
  return added;
}

/* the same rule, batch form (see the header comment) */
typedef struct { int64_t key; size_t i; } orc_key_idx;
static int cmp_key_idx(const void* a, const void* b) {
  const orc_key_idx *x = (const orc_key_idx*)a, *y = (const orc_key_idx*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}
static int cmp_size_t(const void* a, const void* b) {
  const size_t x = *(const size_t*)a, y = *(const size_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

long orc_map_add_points(orc_map* m, const float* in_xyzw, size_t n, const float pose[16]) {
  float* moved = (float*)malloc((n ? n : 1) * 4 * sizeof(float));
  orc_key_idx* ki = (orc_key_idx*)malloc((n ? n : 1) * sizeof(orc_key_idx));
  size_t* first = (size_t*)malloc((n ? n : 1) * sizeof(size_t));
  if (!moved || !ki || !first) {
    free(moved);
    free(ki);
    free(first);
    return -1;
  }
  if (pose) orc_transform_cloud(in_xyzw, n, pose, moved);
  else {
    memcpy(moved, in_xyzw, n * 4 * sizeof(float));
    for (size_t i = 0; i < n; ++i) moved[4 * i + 3] = 1.0f;
  }
  size_t nk = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* p = moved + 4 * i;
    if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
    if (!m->anchored) {
      orc_octree_first_box(p, m->res, &m->ox, &m->oy, &m->oz);
      m->anchored = 1;
    }
    int64_t key;
    if (!point_key(m, p, &key)) continue;
    ki[nk].key = key;
    ki[nk].i = i;
    nk++;
  }
  qsort(ki, nk, sizeof(orc_key_idx), cmp_key_idx);
  /* the first point (lowest index) of every voxel the map does not hold yet */
  size_t nf = 0;
  for (size_t a = 0; a < nk; ++a) {
    if (a && ki[a].key == ki[a - 1].key) continue;
    if (find_key(m->keys, m->n, ki[a].key) >= 0) continue; /* isVoxelOccupiedAtPoint */
    first[nf] = ki[a].i;
    ki[nf].key = ki[a].key; /* (nf <= a: the new voxels' keys, ascending, compacted to the front of ki) */
    nf++;
  }
  qsort(first, nf, sizeof(size_t), cmp_size_t); /* map_cloud_ keeps insertion order */
  const size_t total = m->n + nf;
  if (total > m->cap) {
    size_t cap = m->cap ? m->cap : 4096;
    while (cap < total) cap *= 2;
    float* np_ = (float*)realloc(m->pts, cap * 4 * sizeof(float));
    if (np_) m->pts = np_;
    int64_t* nkeys = (int64_t*)realloc(m->keys, cap * sizeof(int64_t));
    if (nkeys) m->keys = nkeys;
    if (!np_ || !nkeys) {
      free(moved);
      free(ki);
      free(first);
      return -1;
    }
    m->cap = cap;
  }
  for (size_t a = 0; a < nf; ++a) memcpy(m->pts + 4 * (m->n + a), moved + 4 * first[a], 4 * sizeof(float));
  /* merge the new voxels' keys (ki[0 .. nf), ascending) into the sorted key array, from the back */
  {
    size_t w = total, o = m->n;
    for (size_t a = nf; a-- > 0;) {
      while (o > 0 && m->keys[o - 1] > ki[a].key) m->keys[--w] = m->keys[--o];
      m->keys[--w] = ki[a].key;
    }
  }
  m->n = total;
  free(moved);
  free(ki);
  free(first);
  return (long)nf;
}

/* nn cloud of `cloud` seen from `pose`, moved back by pose_inv: out must hold n points; returns the number written */
long orc_map_nn_cloud(const orc_map* m, const float* cloud_xyzw, size_t n, const float pose[16], const float pose_inv[16],
                      float* out_xyzw) {
  if (m->n == 0 || n == 0) return 0;
  int32_t* idx = (int32_t*)malloc(n * sizeof(int32_t));
  float* d2 = (float*)malloc(n * sizeof(float));
  float* sel = (float*)malloc(n * 4 * sizeof(float));
  if (!idx || !d2 || !sel) {
    free(idx);
    free(d2);
    free(sel);
    return -1;
  }
  orc_nn(cloud_xyzw, n, m->pts, m->n, pose, ORC_NN_KDTREE, ORC_ARITH_FMA, idx, d2);
  long k = 0;
  for (size_t i = 0; i < n; ++i)
    if (idx[i] >= 0) memcpy(sel + 4 * (k++), m->pts + 4 * (size_t)idx[i], 4 * sizeof(float));
  orc_transform_cloud(sel, (size_t)k, pose_inv, out_xyzw);
  free(idx);
  free(d2);
  free(sel);
  return k;
}
