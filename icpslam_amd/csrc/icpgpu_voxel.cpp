// icpgpu_voxel.cpp -- host side of the voxel-grid filter (pcl::VoxelGrid as called at icp_odometer.cpp:96-101; kernels: icp_voxel.hip).
#include "icp_ctx.h"


namespace icpgpu_impl {

// pcl::VoxelGrid<PointXYZ>::filter on a device-resident cloud (icp_odometer.cpp:96-101). out receives *n_out points
// (ascending cell index). *passthrough = PCL's "leaf size too small for the input dataset" case: input returned as is.
// publish (icpgpu_voxel_grid_view): the result also goes into the pinned staging buffer, written by a kernel queued right behind the
// filter's last one, and its content fingerprint rides on the read-back of the cell count -- *published says whether that happened
// (the direct path ran and the result fits the staging buffer); *fp_sum is the fingerprint's sum then.
int voxel_filter_device(icpgpu_ctx* c, const float4* d_in, int n, float leaf, DeviceBuf& out, int* n_out, bool* passthrough,
                        int* bbox_enc_out, bool publish, bool* published, unsigned long long* fp_sum) {
  if (published) *published = false;
  *n_out = 0;
  *passthrough = false;
  if (bbox_enc_out)
    for (int a = 0; a < 3; ++a) bbox_enc_out[a] = 1, bbox_enc_out[3 + a] = 0;  // (an empty box until the input's has been read back)
  if (!(leaf > 0.f) || !std::isfinite(leaf)) return fail(c, ICPGPU_ERR_INVALID_ARG, "voxel filter: leaf size must be positive");
  if (n <= 0) return ICPGPU_OK;
  // device ints: [0..5] the encoded bounding box, [6, 7] the cell counts, [8] the direct path's status, [10, 11] the published
  // cloud's fingerprint, [16..23] the plan the device derives from the box (launch_voxel_grid_direct), [24..29] the box as the plan
  // kernel hands it on -- it leaves [0..5] initialised, so the NEXT call's bounding-box pass goes without its init launch
  // (vox_bbox_ready: only after a call that got as far as the plan kernel; anything else starts with the init launch again)
  int rc = ensure(c, c->vox_ints, 32 * sizeof(int));
  if (rc) return rc;
  int* d_ints = static_cast<int*>(c->vox_ints.ptr);
  const bool box_ready = c->vox_bbox_ready == d_ints;
  c->vox_bbox_ready = nullptr;
  HIP_TRY(c, launch_bbox(d_in, n, d_ints, c->stream, !box_ready));
  const float inv = 1.0f / leaf;  // PCL: inverse_leaf_size_ = 1 / leaf_size_ in float
  float ms = 0.f;
  // the direct path (one distribution pass + a sort in LDS): every cloud up to 2M points; it reports the rare cloud it cannot
  // take (thousands of points in one voxel) through `status`, and the library-sort path runs instead
  static const bool force_sort = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_VOXEL_SORT"); return e && std::atoi(e) != 0; }();
  // Round 6: the direct path is queued BEHIND the bounding-box pass without the host having seen the box -- the device derives the
  // filter's parameters itself (voxel_plan_kernel) -- so a scan's filter is ONE wait for the device (box, plan, counts, status and
  // the published cloud's fingerprint arrive together) where it was two.  The rare clouds that are not the direct path's (PCL's
  // pass-through, an index that may wrap, no finite point) show in the plan; the host then goes the old way with the box it has by
  // then.  (development flavour, ICPGPU_VOXEL_PLANNED=0: the box first, as until round 5)
  static const bool planned_enabled = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_VOXEL_PLANNED"); return !e || std::atoi(e) != 0; }();
  const bool direct_size = !force_sort && n <= (1 << 21);
  auto direct_scratch = [&]() -> int {
    int r;
    if ((r = ensure(c, out, (size_t)n * sizeof(float4)))) return r;
    if ((r = ensure(c, c->vox_keys, (size_t)2 * n * sizeof(int)))) return r;
    if ((r = ensure(c, c->vox_vals, (size_t)2 * n * sizeof(int)))) return r;
    if ((r = ensure(c, c->vox_bins, voxel_direct_scratch_ints(n) * sizeof(int)))) return r;
    if (c->vox_bins_zeroed != c->vox_bins.ptr || c->vox_bins_zeroed_cap != c->vox_bins.cap) {
      HIP_TRY(c, hipMemsetAsync(c->vox_bins.ptr, 0, c->vox_bins.cap, c->stream));
      c->vox_bins_zeroed = c->vox_bins.ptr;
      c->vox_bins_zeroed_cap = c->vox_bins.cap;
    }
    if ((r = ensure(c, c->vox_pub, (size_t)voxel_direct_groups(n) * sizeof(unsigned long long)))) return r;
    if (c->vox_pub_zeroed != c->vox_pub.ptr || c->vox_pub_zeroed_cap != c->vox_pub.cap) {
      HIP_TRY(c, hipMemsetAsync(c->vox_pub.ptr, 0, c->vox_pub.cap, c->stream));
      c->vox_pub_zeroed = c->vox_pub.ptr;
      c->vox_pub_zeroed_cap = c->vox_pub.cap;
    }
    return ICPGPU_OK;
  };
  int stage_points = 0;
  auto* d_fp = reinterpret_cast<unsigned long long*>(d_ints + 10);  // (8-byte aligned: the buffer is)
  // queue the direct path's kernels (minb / divb: the host's values, or null = the device's plan)
  auto direct_launch = [&](const int* minb, const int* divb) -> int {
    // (the staging buffer is sized for the input: a filter's result is never longer)
    stage_points = 0;
    if (publish && (size_t)n * sizeof(float4) <= kStageMaxBytes && !ensure_stage(c, (size_t)n * sizeof(float4))) stage_points = n;
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    hipError_t le = launch_voxel_grid_direct(d_in, n, inv, minb, divb, static_cast<int*>(c->vox_bins.ptr),
                                             static_cast<unsigned long long*>(c->vox_pub.ptr),
                                             static_cast<int*>(c->vox_keys.ptr), static_cast<int*>(c->vox_keys.ptr) + n,
                                             static_cast<unsigned long long*>(c->vox_vals.ptr), static_cast<float4*>(out.ptr),
                                             d_ints + 6, d_ints + 8, c->stream, stage_points ? d_fp : nullptr, minb ? nullptr : d_ints,
                                             minb ? nullptr : d_ints + 16);
    if (le != hipSuccess) {
      c->vox_bins_zeroed = nullptr;
      return fail(c, ICPGPU_ERR_HIP, "voxel filter: %s", hipGetErrorString(le));
    }
    HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    if (stage_points)
      HIP_TRY(c, launch_publish_cloud(static_cast<const float4*>(out.ptr), d_ints + 6, n, stage_points, static_cast<float4*>(c->h_stage_dev), d_fp,
                                      c->stream));
    return ICPGPU_OK;
  };
  auto direct_time = [&]() -> int {  // (the events lie in front of the posted kernel: complete by now; should the runtime not have noticed yet, wait for the second)
    hipError_t te = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    if (te == hipErrorNotReady) {
      (void)hipGetLastError();
      HIP_TRY(c, hipEventSynchronize(c->ev[1]));
      te = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    }
    HIP_TRY(c, te);
    return ICPGPU_OK;
  };
  const bool planned = planned_enabled && direct_size;
  int hv[32];
  for (int& v : hv) v = 0;
  if (planned) {
    if ((rc = direct_scratch())) return rc;
    if ((rc = direct_launch(nullptr, nullptr))) return rc;
    if ((rc = fetch_ints(c, d_ints, 30, hv))) {
      c->vox_bins_zeroed = nullptr;
      return rc;
    }
    std::memcpy(hv, hv + 24, 6 * sizeof(int));  // the box, from where the plan kernel put it
    c->vox_bbox_ready = d_ints;                 // ... having left [0..5] initialised
  } else if ((rc = fetch_ints(c, d_ints, 6, hv))) {
    return rc;
  }
  std::memcpy(c->h_ints, hv, 12 * sizeof(int));
  float lo[3], hi[3];
  decode_bbox(hv, lo, hi);
  if (bbox_enc_out) std::memcpy(bbox_enc_out, hv, 6 * sizeof(int));  // the input's box: it contains every centroid
  if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) return ICPGPU_OK;  // no finite point
  int minb[3], divb[3];
  long long d[3];
  for (int a = 0; a < 3; ++a) {
    d[a] = (long long)((hi[a] - lo[a]) * inv) + 1;
    minb[a] = (int)std::floor(lo[a] * inv);
    divb[a] = (int)std::floor(hi[a] * inv) - minb[a] + 1;
  }
  if ((rc = ensure(c, out, (size_t)n * sizeof(float4)))) return rc;
  if (d[0] * d[1] * d[2] > (long long)INT32_MAX) {  // PCL warns and returns the input unchanged
    HIP_TRY(c, hipMemcpyAsync(out.ptr, d_in, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_out = n;
    *passthrough = true;
    return ICPGPU_OK;
  }
  if ((rc = ensure(c, c->vox_keys, (size_t)2 * n * sizeof(int)))) return rc;
  if ((rc = ensure(c, c->vox_vals, (size_t)2 * n * sizeof(int)))) return rc;
  bool done = false;
  // (PCL's overflow test uses the float extents, its cell index the integer ones: when these are one cell wider the index of
  // the topmost cells can wrap in int32 and PCL -- and the sort path, on signed keys -- puts them first.  The direct path's
  // buckets assume keys in [0, number of cells): leave that corner to the sort path.)
  const bool keys_may_wrap = (long long)divb[0] * divb[1] * divb[2] > (long long)INT32_MAX;
  if (planned) {
    // the device decided by the same arithmetic: its verdict and the host's must agree
    const int pre = hv[23];
    if ((pre == 0) != !keys_may_wrap) return fail(c, ICPGPU_ERR_HIP, "voxel filter: the device's plan (%d) and the host's disagree (internal error)", pre);
    if (pre == 0) {
      if ((rc = direct_time())) return rc;
      done = hv[8] == 0;
    }
  } else if (direct_size && !keys_may_wrap) {
    if ((rc = direct_scratch())) return rc;
    if ((rc = direct_launch(minb, divb))) return rc;
    if ((rc = fetch_ints(c, d_ints + 6, stage_points ? 6 : 3, hv + 6))) {
      c->vox_bins_zeroed = nullptr;
      return rc;
    }
    std::memcpy(c->h_ints + 6, hv + 6, 6 * sizeof(int));
    if ((rc = direct_time())) return rc;
    done = hv[8] == 0;
  }
  if (done && stage_points && published) {
    *published = true;
    std::memcpy(fp_sum, hv + 10, sizeof *fp_sum);
  }
  if (!done) {
    const size_t tb = voxel_temp_bytes(n);
    if ((rc = ensure(c, c->vox_flags, (size_t)n * sizeof(int)))) return rc;
    if ((rc = ensure(c, c->vox_slots, (size_t)n * sizeof(int)))) return rc;
    if ((rc = ensure(c, c->vox_temp, tb))) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(c, launch_voxel_grid(d_in, n, inv, minb, divb, static_cast<int*>(c->vox_keys.ptr), static_cast<int*>(c->vox_vals.ptr),
                                 static_cast<int*>(c->vox_flags.ptr), static_cast<int*>(c->vox_slots.ptr), c->vox_temp.ptr, tb,
                                 static_cast<float4*>(out.ptr), d_ints + 6, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_ints + 6, d_ints + 6, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms2 = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms2, c->ev[0], c->ev[1]));
    ms += ms2;
  }
  *n_out = c->h_ints[6] + c->h_ints[7];
  if (*n_out < 0 || *n_out > n) {
    const int got = *n_out;
    *n_out = 0;
    return fail(c, ICPGPU_ERR_HIP, "voxel filter: %d cells from %d points (internal error)", got, n);
  }
  c->prof.voxel_launches += 1;
  c->prof.voxel_ms += ms;
  c->prof.voxel_bytes += 16ull * (uint64_t)n + 16ull * (uint64_t)*n_out;
  return ICPGPU_OK;
}

// The filtered cloud (c->vox_out, m points) to the host.  The reference hands this very cloud back as the registration's
// source a few lines later (icp_odometer.cpp:177 -> :193), so the copy is bracketed by what icpgpu_set_source needs to
// recognise it: the content fingerprint, computed on the device while the copy is in flight (8 more bytes in the same
// synchronisation), and the sample fingerprint of the fetched bytes (a microsecond).
static int fetch_filtered(icpgpu_ctx* c, float* out_xyzw, size_t m) {
  int rc = ensure(c, c->fp_acc, sizeof(unsigned long long));
  if (rc) return rc;
  auto* d_acc = static_cast<unsigned long long*>(c->fp_acc.ptr);
  HIP_TRY(c, launch_fingerprint(static_cast<const float4*>(c->vox_out.ptr), (int)m, d_acc, c->stream));
  int words[2] = {0, 0};  // the 64-bit sum rides on the copy's marker
  if ((rc = copy_to_host(c, out_xyzw, c->vox_out.ptr, m * sizeof(float4), reinterpret_cast<const int*>(d_acc), 2, words))) return rc;
  unsigned long long sum = 0;
  std::memcpy(&sum, words, sizeof sum);
  c->vox_fp = fp_finish(sum, (unsigned long long)m);
  c->vox_sample_fp = sample_fingerprint(out_xyzw, m);
  c->vox_fp_valid = true;
  return ICPGPU_OK;
}

}  // namespace icpgpu_impl

extern "C" {

int icpgpu_voxel_grid(icpgpu_ctx* c, const float* xyzw, size_t n, float leaf, float* out_xyzw, size_t* n_out) {
  ENTER(c);
  if (!n_out || (n && !xyzw)) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  *n_out = 0;
  c->vox_last_n = 0;
  c->vox_fp_valid = c->vox_box_valid = false;
  int rc = ensure(c, c->vox_in, n * sizeof(float4));
  if (rc) return rc;
  if (n) HIP_TRY(c, hipMemcpyAsync(c->vox_in.ptr, xyzw, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  int m = 0;
  bool pass = false;
  int box[6];
  if ((rc = voxel_filter_device(c, static_cast<const float4*>(c->vox_in.ptr), (int)n, leaf, c->vox_out, &m, &pass, box))) return rc;
  c->vox_last_n = (size_t)m;  // out_xyzw == NULL: the filtered cloud waits in HBM for icpgpu_voxel_grid_fetch
  if (m > 0) {
    std::memcpy(c->vox_box, box, sizeof(box));
    c->vox_box_valid = true;
  }
  if (m && out_xyzw && (rc = fetch_filtered(c, out_xyzw, (size_t)m))) return rc;
  *n_out = (size_t)m;
  return ICPGPU_OK;
}

int icpgpu_voxel_grid_view(icpgpu_ctx* c, const float* xyzw, size_t n, float leaf, const float** view_xyzw, size_t* n_out) {
  ENTER(c);
  if (!n_out || !view_xyzw || (n && !xyzw)) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  *n_out = 0;
  *view_xyzw = nullptr;
  c->vox_last_n = 0;
  c->vox_fp_valid = c->vox_box_valid = false;
  int rc = ensure(c, c->vox_in, n * sizeof(float4));
  if (rc) return rc;
  if (n) HIP_TRY(c, hipMemcpyAsync(c->vox_in.ptr, xyzw, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  int m = 0;
  bool pass = false, published = false;
  unsigned long long fp_sum = 0;
  int box[6];
  if ((rc = voxel_filter_device(c, static_cast<const float4*>(c->vox_in.ptr), (int)n, leaf, c->vox_out, &m, &pass, box, /*publish=*/true, &published,
                                &fp_sum)))
    return rc;
  c->vox_last_n = (size_t)m;
  if (m > 0) {
    std::memcpy(c->vox_box, box, sizeof(box));
    c->vox_box_valid = true;
    if (published) {  // the points are in the staging buffer already: they arrived in front of the cell count
      c->vox_fp = fp_finish(fp_sum, (unsigned long long)m);
      c->vox_sample_fp = sample_fingerprint(static_cast<const float*>(c->h_stage), (size_t)m);
      c->vox_fp_valid = true;
      c->prof.voxel_views_direct += 1;
    } else {  // the sort path, a pass-through or a cloud beyond the staging buffer: the copy engine brings them
      if ((rc = ensure_stage(c, (size_t)m * sizeof(float4), /*any_size=*/true))) return rc;
      if ((rc = fetch_filtered(c, static_cast<float*>(c->h_stage), (size_t)m))) return rc;
    }
    *view_xyzw = static_cast<const float*>(c->h_stage);
  }
  *n_out = (size_t)m;
  return ICPGPU_OK;
}

int icpgpu_voxel_grid_fetch(icpgpu_ctx* c, float* out_xyzw, size_t capacity, size_t* n_out) {
  ENTER(c);
  const size_t m = c->vox_last_n;
  if (n_out) *n_out = m;
  if (m > capacity) return fail(c, ICPGPU_ERR_INVALID_ARG, "voxel_grid_fetch: %zu points, room for %zu", m, capacity);
  if (m && !out_xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (m) return fetch_filtered(c, out_xyzw, m);
  return ICPGPU_OK;
}

int icpgpu_set_source_voxel_filtered(icpgpu_ctx* c, const float* xyzw, size_t n, float leaf, size_t* n_out) {
  ENTER(c);
  if (n && !xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null cloud pointer with n = %zu", n);
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  int rc = ensure(c, c->vox_in, n * sizeof(float4));
  if (rc) return rc;
  if (n) HIP_TRY(c, hipMemcpyAsync(c->vox_in.ptr, xyzw, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  int m = 0;
  bool pass = false;
  if (c->src.buf.external) c->src.buf = DeviceBuf{};
  c->src_version++;
  c->src.bbox_version = 0;
  int box[6];
  if ((rc = voxel_filter_device(c, static_cast<const float4*>(c->vox_in.ptr), (int)n, leaf, c->src.buf, &m, &pass, box))) return rc;
  c->src.n = (size_t)m;
  c->src.set = true;
  // The raw scan's bounding box contains every centroid (a mean of points inside a box lies inside it, up to a rounding the
  // grid's one-cell margin absorbs): the grids built over the filtered cloud start from it, without a pass and a round trip
  // of their own.  (An empty / non-finite input leaves lo > hi: a grid build then finds "no finite point", as its own pass would.)
  if (m > 0) {
    std::memcpy(c->src.bbox_enc, box, sizeof(box));
    c->src.bbox_version = c->src_version;
    c->src.bbox_exact = false;
  }
  c->src.sample_valid = false;  // written on the device: no host sample to compare with
  if (n_out) *n_out = (size_t)m;
  return ICPGPU_OK;
}

}  // extern "C"
