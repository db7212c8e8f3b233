// icpgpu_map.cpp -- the mapper's map as ICP target (octree_mapper.cpp:55-90,133-172; kernels: icp_map.hip).
#include "icp_ctx.h"


namespace icpgpu_impl {

int grow_preserving(icpgpu_ctx* c, DeviceBuf& b, size_t keep_bytes, size_t want_bytes) {
  if (want_bytes <= b.cap) return ICPGPU_OK;
  size_t cap = std::max<size_t>(2 * want_bytes, 1u << 20);
  void* np = nullptr;
  HIP_TRY(c, hipMalloc(&np, cap));
  if (keep_bytes && b.ptr) {
    const hipError_t e = hipMemcpyAsync(np, b.ptr, keep_bytes, hipMemcpyDeviceToDevice, c->stream);
    const hipError_t e2 = e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
    if (e2 != hipSuccess) {
      (void)hipFree(np);
      return fail(c, ICPGPU_ERR_HIP, "map growth: %s", hipGetErrorString(e2));
    }
  }
  if (b.ptr && !b.external) HIP_TRY(c, hipFree(b.ptr));
  b.ptr = np;
  b.cap = cap;
  b.external = false;
  return ICPGPU_OK;
}

// host copy of the device transform (same expression, same fused multiply-adds: icp_device.h xform_point)
void xform_point_host(const Xform& T, const float* s, float p[3]) {
  p[0] = std::fmaf(T.m[2], s[2], std::fmaf(T.m[1], s[1], std::fmaf(T.m[0], s[0], T.m[3])));
  p[1] = std::fmaf(T.m[6], s[2], std::fmaf(T.m[5], s[1], std::fmaf(T.m[4], s[0], T.m[7])));
  p[2] = std::fmaf(T.m[10], s[2], std::fmaf(T.m[9], s[1], std::fmaf(T.m[8], s[0], T.m[11])));
}

// The octree's FIRST bounding box, as PCL builds it for the first point ever added (OctreePointCloud::adoptBoundingBoxToPoint on
// an empty octree: box = p +- resolution / 2, then getKeyBitSize()): getKeyBitSize takes max_voxels = max(ceil(extent / res), 2),
// so the tree starts ONE level deep with a side of 2 voxels, and -- the octree holding no leaf yet -- splits the oversize
// evenly: min -= (side - extent) / 2, max += the same.  The box is therefore p +- resolution and the leaf lattice's origin
// p - resolution (until round 4 this code stopped at p +- resolution / 2, depth 0: a lattice shifted by half a voxel).
// Every operation below is PCL's, in double, in PCL's order, so that a resolution that is not a power of two rounds alike.
void pcl_first_box(const float p[3], double res, double bmin[3], double bmax[3], int* depth) {
  const double eps = (double)FLT_EPSILON;  // PCL: const float minValue = std::numeric_limits<float>::epsilon()
  unsigned int max_key = 0;
  for (int a = 0; a < 3; ++a) {
    bmin[a] = (double)p[a] - res / 2;
    bmax[a] = (double)p[a] + res / 2;
    max_key = std::max(max_key, (unsigned int)std::ceil((bmax[a] - bmin[a] - eps) / res));
  }
  const unsigned int max_voxels = std::max(max_key, 2u);
  const unsigned int d = (unsigned int)std::ceil(std::log((double)max_voxels) / std::log(2.0) - eps);  // Log2(n) = log(n) / log(2)
  const double side = (double)(1u << d) * res;
  for (int a = 0; a < 3; ++a) {
    const double oversize = (side - (bmax[a] - bmin[a])) / 2.0;
    if (oversize > eps) {
      bmin[a] -= oversize;
      bmax[a] += oversize;
    }
  }
  *depth = (int)d;
}

// The lattice is anchored by the first point that is ever added: origin = the minimum corner of that first box (the box only
// ever grows by whole octree side lengths afterwards, so the lattice never moves).  d_in: the batch in HBM.
int map_anchor(icpgpu_ctx* c, const float4* d_in, int n, const Xform& T) {
  VoxelMap& M = c->map;
  std::vector<float> chunk;
  for (int off = 0; off < n && !M.anchored; off += 4096) {
    const int m = std::min(4096, n - off);
    chunk.resize((size_t)m * 4);
    HIP_TRY(c, hipMemcpyAsync(chunk.data(), d_in + off, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < m; ++i) {
      float p[3];
      xform_point_host(T, &chunk[(size_t)i * 4], p);
      if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])) {
        double bmin[3], bmax[3];
        int depth;
        pcl_first_box(p, M.desc.res, bmin, bmax, &depth);
        M.desc.ox = bmin[0];
        M.desc.oy = bmin[1];
        M.desc.oz = bmin[2];
        M.anchored = true;
        break;
      }
    }
  }
  return ICPGPU_OK;
}

// addPointsToMap() for a batch already in HBM
int map_insert_device(icpgpu_ctx* c, const float4* d_in, int n, const float* pose, size_t* n_added) {
  VoxelMap& M = c->map;
  if (n_added) *n_added = 0;
  if (!M.defined) return fail(c, ICPGPU_ERR_NO_INPUT, "map: call icpgpu_map_reset first");
  if (n <= 0) return ICPGPU_OK;
  if ((size_t)M.n + (size_t)n > (size_t)INT32_MAX / 4) return fail(c, ICPGPU_ERR_INVALID_ARG, "map too large");
  const Xform T = to_xform(pose);
  int rc;
  if (!M.anchored && (rc = map_anchor(c, d_in, n, T))) return rc;
  if (!M.anchored) return ICPGPU_OK;  // nothing finite in the batch: the map stays empty

  // hash set with load <= 1/2 even if every point of the batch opens a new voxel
  const size_t need = 2 * ((size_t)M.n + (size_t)n);
  if (need > M.cap) {
    unsigned int cap = 1u << 16;
    while (cap < need) cap <<= 1;
    if ((rc = ensure(c, M.keys, (size_t)cap * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, M.vals, (size_t)cap * sizeof(int)))) return rc;
    if ((rc = ensure(c, M.first, (size_t)cap * sizeof(int)))) return rc;
    M.cap = cap;
    HIP_TRY(c, launch_map_fill(static_cast<unsigned long long*>(M.keys.ptr), static_cast<int*>(M.vals.ptr),
                               static_cast<int*>(M.first.ptr), cap, c->stream));
    HIP_TRY(c, launch_map_rehash(M.pts.data(), M.n, M.desc, static_cast<unsigned long long*>(M.keys.ptr),
                                 static_cast<int*>(M.vals.ptr), cap, c->stream));
  }
  if ((rc = grow_preserving(c, M.pts.buf, (size_t)M.n * sizeof(float4), ((size_t)M.n + (size_t)n) * sizeof(float4)))) return rc;
  if ((rc = ensure(c, M.moved, (size_t)n * sizeof(float4)))) return rc;
  if ((rc = ensure(c, M.slot_of, (size_t)n * sizeof(int)))) return rc;
  if ((rc = ensure(c, M.flags, (size_t)n * sizeof(int)))) return rc;
  if ((rc = ensure(c, M.rank, (size_t)n * sizeof(int)))) return rc;
  const size_t temp_bytes = map_scan_temp_bytes(n);
  if ((rc = ensure(c, M.temp, temp_bytes))) return rc;
  if ((rc = ensure(c, M.counter, 4 * sizeof(int)))) return rc;
  int* d_added = static_cast<int*>(M.counter.ptr);
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, launch_map_insert(d_in, n, T, M.desc, static_cast<unsigned long long*>(M.keys.ptr), static_cast<int*>(M.vals.ptr),
                               static_cast<int*>(M.first.ptr), M.cap, static_cast<float4*>(M.moved.ptr),
                               static_cast<int*>(M.slot_of.ptr), static_cast<int*>(M.flags.ptr), static_cast<int*>(M.rank.ptr),
                               M.temp.ptr, temp_bytes, M.n, static_cast<float4*>(M.pts.buf.ptr), d_added, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_added, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  const int added = c->h_ints[0];
  M.n += added;
  M.pts.n = (size_t)M.n;
  M.pts.set = true;
  if (added > 0) M.version++;
  c->prof.map_inserts += 1;
  c->prof.map_insert_ms += ms;
  c->prof.map_points_in += (uint64_t)n;
  if (n_added) *n_added = (size_t)added;
  return ICPGPU_OK;
}


// OctreePointCloud::adoptBoundingBoxToPoint for one point outside the box (oracle/map_approx_np.py::_adopt restates the same):
// the box doubles -- all three axes at once, the old root becoming the UPPER child on every axis the point does not violate
// from above -- until the point is inside
void approx_adopt(VoxelMap& M, const float p[3]) {
  ApproxBox& b = M.box;
  const double eps = (double)FLT_EPSILON;
  for (;;) {
    if (!M.box_defined) {
      pcl_first_box(p, b.res, b.min, b.max, &b.depth);  // p +- res, depth 1 (getKeyBitSize)
      M.box_defined = true;
      continue;
    }
    bool lower[3], upper[3], any = false;
    for (int a = 0; a < 3; ++a) {
      lower[a] = (double)p[a] < b.min[a];
      upper[a] = (double)p[a] >= b.max[a];
      any = any || lower[a] || upper[a];
    }
    if (!any) return;
    const double side = (double)(1ll << b.depth) * b.res;
    for (int a = 0; a < 3; ++a)
      if (!upper[a]) {
        b.min[a] -= side;
        M.box_shift[a] += 1ll << b.depth;  // stored keys move by whole voxels with the minimum
      }
    b.depth += 1;
    for (int a = 0; a < 3; ++a) b.max[a] = b.min[a] + ((double)(1ll << b.depth) * b.res - eps);
  }
}

// bring the box and the node set up to date with the map, then keys[i] = approxNearestSearch(pose * source[i])
int approx_nn_keys(icpgpu_ctx* c, const Xform& T, int n_s, unsigned long long* keys) {
  VoxelMap& M = c->map;
  int rc;
  if ((rc = ensure(c, M.counter, 4 * sizeof(int)))) return rc;
  int* d_first = static_cast<int*>(M.counter.ptr) + 2;
  M.box.res = M.desc.res;
  // (1) the bounding box: replay the growth over the points added since the last call (a handful of round trips per map LIFE)
  while (M.box_upto < M.n) {
    if (M.box_defined) {
      HIP_TRY(c, launch_approx_first_outside(M.pts.data() + M.box_upto, M.n - M.box_upto, M.box, d_first, c->stream));
      HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_first, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (c->h_ints[0] >= M.n - M.box_upto) {  // all inside
        M.box_upto = M.n;
        break;
      }
      M.box_upto += c->h_ints[0];
    }
    float p[4];
    HIP_TRY(c, hipMemcpyAsync(p, M.pts.data() + M.box_upto, sizeof(p), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    approx_adopt(M, p);
    M.box_version++;
    if (M.box_hist.n >= kApproxMaxVersions) return fail(c, ICPGPU_ERR_UNSUPPORTED, "map: the octree's bounding box changed more than %d times", kApproxMaxVersions);
    {  // the version in force from this point on (the point that made the box grow is keyed under the grown box)
      ApproxHistory& H = M.box_hist;
      H.first[H.n] = M.box_upto;
      for (int a = 0; a < 3; ++a) {
        H.min[H.n][a] = M.box.min[a];
        H.shift[H.n][a] = M.box_shift[a];
      }
      H.n += 1;
    }
    M.box_upto += 1;
    if (M.box.depth > approx_max_depth()) break;  // (reported below -- on this call and on every later one)
  }
  // the depth limit holds for EVERY call, not only for the one that grew the box: node keys pack 19 bits per axis
  if (M.box.depth > approx_max_depth())
    return fail(c, ICPGPU_ERR_UNSUPPORTED, "map: the octree is %d levels deep (approximate search supports %d)", M.box.depth,
                approx_max_depth());
  // (2) the set of occupied nodes: level d holds at most min(n, 8^d) of them
  size_t nodes = 0;
  for (int d = 1; d <= M.box.depth; ++d) {
    const double full = std::pow(8.0, (double)d);
    nodes += (size_t)std::min((double)M.n, full);
  }
  unsigned int cap = 1u << 12;
  while ((size_t)cap < 2 * nodes + 16) cap <<= 1;
  bool rebuild = M.nodes_box_version != M.box_version;
  if (cap > M.node_cap) {
    if ((rc = ensure(c, M.node_keys, (size_t)cap * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, M.node_vals, (size_t)cap * sizeof(int)))) return rc;
    M.node_cap = cap;
    rebuild = true;
  }
  auto* nk = static_cast<unsigned long long*>(M.node_keys.ptr);
  auto* nv = static_cast<int*>(M.node_vals.ptr);
  if (rebuild) {
    HIP_TRY(c, launch_approx_fill(nk, nv, M.node_cap, c->stream));
    M.nodes_upto = 0;
    M.nodes_box_version = M.box_version;
  }
  HIP_TRY(c, launch_approx_insert(M.pts.data(), M.nodes_upto, M.n, M.box, M.box_hist, nk, nv, M.node_cap, c->stream));
  M.nodes_upto = M.n;
  // (3) the descent
  HIP_TRY(c, launch_approx_descend(c->src.data(), n_s, T, M.box, nk, nv, M.node_cap, keys, c->stream));
  return ICPGPU_OK;
}

}  // namespace icpgpu_impl

extern "C" {

int icpgpu_map_set_search(icpgpu_ctx* c, int mode) {
  ENTER(c);
  if (mode != ICPGPU_MAP_SEARCH_EXACT && mode != ICPGPU_MAP_SEARCH_PCL_APPROX) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad map search mode");
  c->map.search_mode = mode;
  return ICPGPU_OK;
}

int icpgpu_map_reset(icpgpu_ctx* c, double resolution) {
  ENTER(c);
  if (!(resolution > 0.0) || !std::isfinite(resolution)) return fail(c, ICPGPU_ERR_INVALID_ARG, "map: resolution must be positive");
  VoxelMap& M = c->map;
  M.defined = true;
  M.anchored = false;
  M.desc = MapDesc{0.0, 0.0, 0.0, resolution};
  M.n = 0;
  M.pts.n = 0;
  M.pts.set = true;
  M.version++;
  M.cap = 0;  // the hash set is rebuilt by the next insertion
  M.box_defined = false;
  M.box_hist.n = 0;
  M.box_shift[0] = M.box_shift[1] = M.box_shift[2] = 0;
  M.box_upto = 0;
  M.box_version++;
  M.nodes_upto = 0;
  M.nodes_box_version = ~0ull;
  M.grid.built = M.grid.usable = false;
  return ICPGPU_OK;
}

int icpgpu_map_add_points(icpgpu_ctx* c, const float* xyzw, size_t n, const float* pose, size_t* n_added) {
  ENTER(c);
  if (n_added) *n_added = 0;
  if (n && !xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null cloud pointer with n = %zu", n);
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  if (!c->map.defined) return fail(c, ICPGPU_ERR_NO_INPUT, "map: call icpgpu_map_reset first");
  if (n == 0) return ICPGPU_OK;
  int rc = ensure(c, c->map.staged, n * sizeof(float4));
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->map.staged.ptr, xyzw, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may free xyzw as soon as we return
  return map_insert_device(c, static_cast<const float4*>(c->map.staged.ptr), (int)n, pose, n_added);
}

int icpgpu_map_add_source(icpgpu_ctx* c, const float* pose, size_t* n_added) {
  ENTER(c);
  if (n_added) *n_added = 0;
  if (!c->src.set) return fail(c, ICPGPU_ERR_NO_INPUT, "map_add_source: no source set");
  return map_insert_device(c, c->src.data(), (int)c->src.n, pose, n_added);
}

int icpgpu_map_size(icpgpu_ctx* c, size_t* n) {
  if (!c || !n) return ICPGPU_ERR_INVALID_ARG;
  *n = (size_t)c->map.n;
  return ICPGPU_OK;
}

int icpgpu_map_get_points(icpgpu_ctx* c, float* out_xyzw, size_t capacity, size_t* n) {
  ENTER(c);
  if (n) *n = (size_t)c->map.n;
  if ((size_t)c->map.n > capacity) return fail(c, ICPGPU_ERR_INVALID_ARG, "map_get_points: %d points, room for %zu", c->map.n, capacity);
  if (c->map.n > 0) {
    if (!out_xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null output");
    HIP_TRY(c, hipMemcpyAsync(out_xyzw, c->map.pts.buf.ptr, (size_t)c->map.n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return ICPGPU_OK;
}

int icpgpu_map_nn_target(icpgpu_ctx* c, const float* pose, const float* pose_inv, float* nn_out_xyzw, size_t* n_nn) {
  ENTER(c);
  if (n_nn) *n_nn = 0;
  VoxelMap& M = c->map;
  if (!M.defined) return fail(c, ICPGPU_ERR_NO_INPUT, "map: call icpgpu_map_reset first");
  if (!c->src.set) return fail(c, ICPGPU_ERR_NO_INPUT, "map_nn_target: no source set");
  const int n_s = (int)c->src.n;
  if (c->tgt.buf.external) c->tgt.buf = DeviceBuf{};
  c->tgt_version++;
  c->tgt.set = true;
  c->tgt.n = 0;
  c->tgt.bbox_version = 0;
  c->tgt.sample_valid = false;
  c->have_final = false;
  if (M.n == 0 || n_s == 0) return ICPGPU_OK;  // approxNearestNeighbors() on an empty map: empty nn cloud

  // exact NN of pose * s in the map: grid where the neighbour is within 2 voxel sizes, brute force for the rest
  int rc = ICPGPU_OK;
  if (M.search_mode == ICPGPU_MAP_SEARCH_EXACT && (rc = build_grid(c, M.pts, M.version, 2.0 * M.desc.res, /*adapt=*/true, M.grid))) return rc;
  if ((rc = ensure(c, M.nn_keys, (size_t)n_s * sizeof(unsigned long long)))) return rc;
  auto* keys = static_cast<unsigned long long*>(M.nn_keys.ptr);
  const Xform T = to_xform(pose), Tinv = to_xform(pose_inv);
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (M.search_mode == ICPGPU_MAP_SEARCH_PCL_APPROX) {
    if ((rc = approx_nn_keys(c, T, n_s, keys))) return rc;
  } else if (M.grid.usable) {
    if ((rc = nn_keys_grid(c, M.grid, c->src.data(), n_s, M.pts.data(), M.n, T, keys))) return rc;
  } else {
    if ((rc = nn_keys_brute(c, M.pts.data(), M.n, T, keys))) return rc;
  }
  if ((rc = ensure(c, c->tgt.buf, (size_t)n_s * sizeof(float4)))) return rc;
  if ((rc = ensure(c, M.flags, (size_t)n_s * sizeof(int)))) return rc;
  if ((rc = ensure(c, M.rank, (size_t)n_s * sizeof(int)))) return rc;
  const size_t temp_bytes = map_scan_temp_bytes(n_s);
  if ((rc = ensure(c, M.temp, temp_bytes))) return rc;
  if ((rc = ensure(c, M.counter, 4 * sizeof(int)))) return rc;
  int* d_count = static_cast<int*>(M.counter.ptr);
  HIP_TRY(c, launch_map_nn_gather(keys, n_s, M.pts.data(), Tinv, static_cast<int*>(M.flags.ptr), static_cast<int*>(M.rank.ptr),
                                  M.temp.ptr, temp_bytes, static_cast<float4*>(c->tgt.buf.ptr), d_count, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  c->prof.map_nn_launches += 1;
  c->prof.map_nn_ms += ms;
  const int m = c->h_ints[0];
  c->tgt.n = (size_t)m;
  c->tgt.sample_valid = false;  // written on the device: no host sample to compare with
  // The nn cloud repeats every chosen map point ~30x (0.5 m voxels, 200k-point scans).  Repeats add nothing to a nearest-
  // neighbour search but make every cell of the target's grid 30x denser, so the grid for the coming align is built here
  // from the DISTINCT points, each carrying the index of its first occurrence in the nn cloud -- exactly the index the
  // lowest-index tie-break would report on the full cloud.  (A later change of the correspondence gate simply rebuilds
  // the grid from the full cloud.)
  const int mode = c->params.nn_mode;
  const float thr = threshold_from(c->params.max_correspondence_distance * c->params.max_correspondence_distance);
  const double cut = std::sqrt((double)thr) * (1.0 + 1e-6);
  if (m > 0 && (mode == ICPGPU_NN_GRID || (mode == ICPGPU_NN_AUTO && (size_t)m >= kGridMinTarget)) && thr > 0.f && std::isfinite(cut) &&
      cut <= 1e6) {
    if ((rc = ensure(c, M.first_user, (size_t)M.n * sizeof(int)))) return rc;
    if ((rc = ensure(c, M.uflags, (size_t)n_s * sizeof(int)))) return rc;
    if ((rc = ensure(c, M.urank, (size_t)n_s * sizeof(int)))) return rc;
    if ((rc = ensure(c, M.uniq_index, (size_t)n_s * sizeof(int)))) return rc;
    if ((rc = ensure(c, M.uniq.buf, (size_t)n_s * sizeof(float4)))) return rc;
    HIP_TRY(c, launch_map_nn_unique(keys, static_cast<const int*>(M.flags.ptr), static_cast<const int*>(M.rank.ptr), n_s,
                                    static_cast<const float4*>(c->tgt.buf.ptr), M.n, static_cast<int*>(M.first_user.ptr),
                                    static_cast<int*>(M.uflags.ptr), static_cast<int*>(M.urank.ptr), M.temp.ptr, temp_bytes,
                                    static_cast<float4*>(M.uniq.buf.ptr), static_cast<int*>(M.uniq_index.ptr), d_count + 1, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_count + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    M.uniq.n = (size_t)c->h_ints[0];
    M.uniq.set = true;
    c->grid.built = false;
    if ((rc = build_grid(c, M.uniq, c->tgt_version, cut, /*adapt=*/true, c->grid, static_cast<const int*>(M.uniq_index.ptr)))) return rc;
  }
  if (nn_out_xyzw && m > 0) {
    if ((rc = copy_to_host(c, nn_out_xyzw, c->tgt.buf.ptr, (size_t)m * sizeof(float4)))) return rc;
  }
  if (n_nn) *n_nn = (size_t)m;
  return ICPGPU_OK;
}

}  // extern "C"
