// icpgpu_context.cpp -- the C-ABI of libicpgpu.so (include/icpgpu.h): context life cycle, parameters, clouds, profile.
//
// Host side of the hot path behind the PCL Registration protocol used at
//   /root/reference/src/icpslam/icp_odometer.cpp:188-201 and src/icpslam/octree_mapper.cpp:104-117.
// There is deliberately no CPU fallback: if HIP or the device is unusable every entry point fails loudly.
#include "icp_ctx.h"

#include <mutex>


namespace icpgpu_impl {

thread_local std::string g_create_error;  // icpgpu_create failures (no context to carry the message yet)

int fail(icpgpu_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return code;
}


extern std::atomic<unsigned long long> g_alloc_calls, g_alloc_us;
int ensure(icpgpu_ctx* c, DeviceBuf& b, size_t bytes) {
  if (b.external) {
    b.ptr = nullptr;
    b.cap = 0;
    b.external = false;
  }
  if (bytes <= b.cap) return ICPGPU_OK;
  {
    static const bool trace = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_ALLOC_TRACE"); return e && std::atoi(e) != 0; }();
    if (trace) fprintf(stderr, "[icpgpu] alloc: %zu -> %zu bytes (context %p)\n", b.cap, bytes, (void*)c);
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (b.ptr) HIP_TRY(c, hipFree(b.ptr));
  b.ptr = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4;  // amortise growth across scans of slightly different size
  if (want < 256) want = 256;
  HIP_TRY(c, hipMalloc(&b.ptr, want));
  b.cap = want;
  g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
  g_alloc_us.fetch_add((unsigned long long)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
  return ICPGPU_OK;
}
// (process-wide: device allocations made through ensure() and the host microseconds they took -- hipFree synchronises the device;
//  read by the batch scheduler's trace, ICPGPU_BATCH_TRACE in the development flavour)
std::atomic<unsigned long long> g_alloc_calls{0}, g_alloc_us{0};

void release(DeviceBuf& b) {
  if (b.ptr && !b.external) (void)hipFree(b.ptr);
  b = DeviceBuf{};
}

Xform to_xform(const Mat4d& T) {
  float f[16];
  mat4_to_float(T, f);
  Xform x;
  for (int r = 0; r < 3; ++r) {
    x.m[4 * r + 0] = f[0 * 4 + r];
    x.m[4 * r + 1] = f[1 * 4 + r];
    x.m[4 * r + 2] = f[2 * 4 + r];
    x.m[4 * r + 3] = f[3 * 4 + r];
  }
  return x;
}

Xform to_xform(const float* T) {
  Mat4d m;
  for (int i = 0; i < 16; ++i) m[i] = T ? (double)T[i] : (i % 5 == 0 ? 1.0 : 0.0);
  return to_xform(m);
}

// largest float f with (double)f <= r2, so that the device's float compare equals PCL's double compare
float threshold_from(double r2) {
  if (std::isnan(r2)) return NAN;
  if (r2 >= (double)FLT_MAX) return FLT_MAX;
  if (r2 < 0.0) return -1.0f;
  float f = (float)r2;
  if ((double)f > r2) f = std::nextafterf(f, -INFINITY);
  return f;
}

unsigned long long sample_fingerprint(const float* xyzw, size_t n) {
  unsigned long long s = 0;
  const unsigned char* b = reinterpret_cast<const unsigned char*>(xyzw);
  const size_t m = n < 256 ? n : 256;
  for (size_t k = 0; k < m; ++k) {
    const size_t i = (k * n) / m;
    unsigned long long w0, w1;
    std::memcpy(&w0, b + 16 * i, 8);
    std::memcpy(&w1, b + 16 * i + 8, 8);
    s += fp_point(w0, w1, (unsigned long long)i);
  }
  return fp_finish(s, (unsigned long long)n);
}

// Spin until the n result pairs at `box` (mapped host memory: post_ints_kernel's target) carry `number`, then hand their values out
// (out may be null).  The stream is queried now and then so that a faulted kernel turns into an error instead of an endless wait,
// and the clock so that a hung one does.  `what` names the thing waited for in those messages.
int wait_posted(icpgpu_ctx* c, const volatile unsigned long long* box, int n, unsigned long long number, int* out, const char* what) {
  std::chrono::steady_clock::time_point t0;
  for (unsigned spins = 1;; ++spins) {
    bool all = true;
    for (int k = 0; k < n && all; ++k) all = (box[2 * k + 1] >> 24) == number;
    if (all) {
      unsigned long long bits;
      for (int k = 0; k < n && all; ++k) {
        all = mailbox_read(box + 2 * k, number, &bits);  // (a torn pair: looked at again)
        if (all && out) out[k] = (int)(unsigned int)bits;
      }
      if (all) break;
    }
    if ((spins & 0x3FFu) == 0) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for %s: %s", what, hipGetErrorString(q));
      const auto now = std::chrono::steady_clock::now();
      if (spins == 0x400u) t0 = now;
      else if (std::chrono::duration<double, std::milli>(now - t0).count() > wait_timeout_ms())
        return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for %s (hung kernel?)", wait_timeout_ms(), what);
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ICPGPU_OK;
}

int fetch_ints(icpgpu_ctx* c, const int* d_src, int n, int* host_dst) {
  if (n <= 0) return ICPGPU_OK;
  if (n > 32) return fail(c, ICPGPU_ERR_INVALID_ARG, "fetch_ints: %d values", n);
  const unsigned long long number = ++c->post_seq;
  HIP_TRY(c, launch_post_ints(d_src, n, c->h_post_dev, wire_seq(c, number), c->stream));
  return wait_posted(c, c->h_post, n, number, host_dst, "a device read-back");
}

// Device -> caller's (pageable) host buffer through the context's PINNED staging buffer: the runtime's own pageable copy takes
// ~250 us for the 340 KB of a voxel-filtered scan (measured: the aligned cloud of every icpgpu_align the C++ shim makes, the
// filtered cloud of every VoxelGrid::filter); a DMA into pinned memory, its end seen through a posted marker (fetch_ints: no stream
// synchronisation), and a memcpy take ~60.  `extra_ints` (optional, n_extra <= 8 ints of device memory) ride on the same marker.
// Copies above kStageMaxBytes go the direct way (the staging buffer is not meant to hold a raw 16 MB submap).
int ensure_stage(icpgpu_ctx* c, size_t bytes, bool any_size) {
  if (bytes <= c->h_stage_cap) return ICPGPU_OK;
  if (bytes > kStageMaxBytes && !any_size) return fail(c, ICPGPU_ERR_UNSUPPORTED, "staging buffer: %zu bytes asked for", bytes);
  if (c->h_stage) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // (nothing in flight may still write to the old one)
    (void)hipHostFree(c->h_stage);
  }
  c->h_stage = c->h_stage_dev = nullptr;
  c->h_stage_cap = 0;
  const size_t want = any_size && bytes > kStageMaxBytes ? bytes : std::min(kStageMaxBytes, std::max<size_t>(bytes + bytes / 4, 1u << 20));
  // mapped + coherent: kernels store into it (a kernel's end makes its stores visible to the host before the next kernel's marker)
  HIP_TRY(c, hipHostMalloc(&c->h_stage, want, hipHostMallocMapped | hipHostMallocCoherent));
  hipError_t e = hipHostGetDevicePointer(&c->h_stage_dev, c->h_stage, 0);
  if (e != hipSuccess) {
    (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    return fail(c, ICPGPU_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e));
  }
  c->h_stage_cap = want;
  return ICPGPU_OK;
}

// the staged clouds' markers live behind fetch_ints' 32 pairs and the 8 of a grid's statistics
static constexpr int kStagePostSlot = 40, kStagePostInts = 8;

int stage_post(icpgpu_ctx* c, const int* d_ints, int n_ints, StageTicket& tk) {
  if (n_ints > kStagePostInts) return fail(c, ICPGPU_ERR_INVALID_ARG, "stage_post: %d values", n_ints);
  if (n_ints <= 0) {  // any device int will do for the marker
    int rc = ensure(c, c->fp_acc, sizeof(unsigned long long));
    if (rc) return rc;
    d_ints = static_cast<const int*>(c->fp_acc.ptr);
    n_ints = 1;
  }
  tk.number = ++c->post_seq;
  tk.n_ints = n_ints;
  HIP_TRY(c, launch_post_ints(d_ints, n_ints, c->h_post_dev + 2 * kStagePostSlot, wire_seq(c, tk.number), c->stream));
  tk.issued = true;
  return ICPGPU_OK;
}

int stage_wait(icpgpu_ctx* c, StageTicket& tk, int* ints_out) {
  if (!tk.issued) return fail(c, ICPGPU_ERR_INVALID_ARG, "stage_wait: nothing posted");
  tk.issued = false;
  return wait_posted(c, c->h_post + 2 * kStagePostSlot, tk.n_ints, tk.number, ints_out, "a staged cloud");
}

int copy_to_host(icpgpu_ctx* c, void* dst, const void* d_src, size_t bytes, const int* d_extra, int n_extra, int* extra_out) {
  int rc;
  if (bytes == 0 && n_extra == 0) return ICPGPU_OK;
  if (bytes > kStageMaxBytes) {
    HIP_TRY(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    if (n_extra) HIP_TRY(c, hipMemcpyAsync(c->h_ints + 8, d_extra, (size_t)n_extra * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_extra) std::memcpy(extra_out, c->h_ints + 8, (size_t)n_extra * sizeof(int));
    return ICPGPU_OK;
  }
  if ((rc = ensure_stage(c, bytes))) return rc;
  if (bytes) HIP_TRY(c, hipMemcpyAsync(c->h_stage, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  int dummy = 0;
  if (n_extra == 0) {  // any device int will do for the marker
    if ((rc = ensure(c, c->fp_acc, sizeof(unsigned long long)))) return rc;
    d_extra = static_cast<const int*>(c->fp_acc.ptr);
    n_extra = 1;
    extra_out = &dummy;
  }
  if ((rc = fetch_ints(c, d_extra, n_extra, extra_out))) return rc;  // queued behind the copy: returns when both are there
  if (bytes && dst != c->h_stage) std::memcpy(dst, c->h_stage, bytes);  // (a view's caller reads the staging buffer itself)
  return ICPGPU_OK;
}

int set_cloud_host(icpgpu_ctx* c, Cloud& cl, const float* xyzw, size_t n, bool sync) {
  if (n > 0 && !xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null cloud pointer with n = %zu", n);
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  int rc = ensure(c, cl.buf, n * sizeof(float4));
  if (rc) return rc;
  if (n) {
    HIP_TRY(c, hipMemcpyAsync(cl.buf.ptr, xyzw, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    if (sync) HIP_TRY(c, hipStreamSynchronize(c->stream));  // the caller may free xyzw as soon as we return
  }
  cl.n = n;
  cl.set = true;
  cl.bbox_version = 0;
  cl.sample_valid = n > 0;
  cl.sample_fp = n > 0 ? sample_fingerprint(xyzw, n) : 0;
  return ICPGPU_OK;
}

int set_cloud_device(icpgpu_ctx* c, Cloud& cl, const void* d_xyzw, size_t n) {
  if (n > 0 && !d_xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null device pointer with n = %zu", n);
  if (((uintptr_t)d_xyzw & 15u) != 0) return fail(c, ICPGPU_ERR_INVALID_ARG, "device cloud must be 16-byte aligned");
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  release(cl.buf);
  cl.buf.ptr = const_cast<void*>(d_xyzw);
  cl.buf.external = true;
  cl.n = n;
  cl.set = true;
  cl.bbox_version = 0;
  cl.sample_valid = false;
  return ICPGPU_OK;
}

// ICPGPU_RECOGNISE=0: icpgpu_set_target always uploads (A/B measurements)
static bool recognise_enabled() {
  static const bool v = [] { const char* e = std::getenv("ICPGPU_RECOGNISE"); return !e || std::atoi(e) != 0; }();
  return v;
}

// fingerprint of a cloud in HBM (cached per version): one small kernel + an 8-byte read-back
int device_fingerprint(icpgpu_ctx* c, const Cloud& cl, uint64_t version, unsigned long long& cache, uint64_t& cache_version,
                       unsigned long long* out) {
  if (cache_version != version) {
    int rc = ensure(c, c->fp_acc, sizeof(unsigned long long));
    if (rc) return rc;
    auto* d_acc = static_cast<unsigned long long*>(c->fp_acc.ptr);
    HIP_TRY(c, launch_fingerprint(cl.data(), (int)cl.n, d_acc, c->stream));
    unsigned long long sum = 0;
    HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_acc, sizeof sum, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memcpy(&sum, c->h_ints, sizeof sum);
    cache = fp_finish(sum, (unsigned long long)cl.n);
    cache_version = version;
  }
  *out = cache;
  return ICPGPU_OK;
}


// GICP's resources -- the evaluation mailbox, the server's command line, the device solver's slots and result granules -- are
// allocated when a context first registers in GICP mode (align_gicp, gicp_run_begin), not with the context: five allocations
// and three synchronous memsets that a point-to-point batch worker never uses (icpgpu_align_batch creates up to 64 workers).
int ensure_gicp_resources(icpgpu_ctx* c) {
  if (c->gicp_resources_ready) return ICPGPU_OK;
  // Every failure path leaves the context as it found it (nothing half-allocated that a second call would allocate over):
  // each buffer is guarded by its own pointer and released again when a later step of its group fails.
  hipError_t e;
  if (!c->h_gicp) {
    const size_t n_flags_end = (size_t)kGicpDirectBlocks * kGicpPartialStride + 8 + kGicpDirectBlocks;  // partials, gap, flags
    const size_t cmd_off = (n_flags_end + 7) & ~(size_t)7;                                        // 64-byte aligned
    const size_t n_d = cmd_off + 8;                                                              // + the server's command line
    double* h = nullptr;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h), n_d * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
      return fail(c, e == hipErrorOutOfMemory ? ICPGPU_ERR_OOM : ICPGPU_ERR_HIP, "hipHostMalloc: %s", hipGetErrorString(e));
    std::memset(h, 0, n_d * sizeof(double));
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_gicp_dev), h, 0)) != hipSuccess) {
      (void)hipHostFree(h);
      return fail(c, ICPGPU_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e));
    }
    c->h_gicp = h;
    const size_t off = (size_t)kGicpDirectBlocks * kGicpPartialStride + 8;
    c->h_gicp_flags = reinterpret_cast<volatile unsigned long long*>(c->h_gicp + off);
    c->h_gicp_flags_dev = reinterpret_cast<unsigned long long*>(c->h_gicp_dev + off);
  }
  if (gicp_device_solver_enabled() && !c->gicp_device_ok) {  // the device solver's slots and result granules; without them GICP solves on the host
    const size_t slot_bytes = gicp_solve_slot_bytes(kGicpDirectBlocks);
    void* h = nullptr;
    unsigned long long* slots = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&slots), slot_bytes, hipDeviceMallocFinegrained) == hipSuccess &&
        hipMemset(slots, 0, slot_bytes) == hipSuccess &&
        hipHostMalloc(&h, 512, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      std::memset(h, 0, 512);
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_solve_dev), h, 0) == hipSuccess) {
        c->gicp_slots = slots;
        c->h_solve = static_cast<volatile unsigned long long*>(h);
        c->gicp_device_ok = true;
      }
    }
    if (!c->gicp_device_ok) {  // no device solver on this context: nothing of it stays behind
      (void)hipGetLastError();
      if (slots) (void)hipFree(slots);
      if (h) (void)hipHostFree(h);
      c->h_solve_dev = nullptr;
    }
    if (c->gicp_device_ok) {
      static std::atomic<int> serial{0};
      c->gicp_xcc = serial.fetch_add(1) & 7;  // concurrent contexts spread their one-XCD runs over the eight XCDs
      const size_t owner_bytes = (size_t)kGicpDirectBlocks * sizeof(unsigned long long);
      unsigned long long *local = nullptr, *owner = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&local), slot_bytes) == hipSuccess && hipMemset(local, 0, slot_bytes) == hipSuccess &&
          hipMalloc(reinterpret_cast<void**>(&owner), owner_bytes) == hipSuccess && hipMemset(owner, 0, owner_bytes) == hipSuccess) {
        c->gicp_slots_local = local;
        c->gicp_owner = owner;
        c->gicp_local_ok = true;
      } else {
        (void)hipGetLastError();
        if (local) (void)hipFree(local);
        if (owner) (void)hipFree(owner);
      }
    }
  }
  if (gicp_server_enabled() && !c->gicp_cmd) {  // no such memory (no large BAR): the evaluations stay single launches
    unsigned int* cmd = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&cmd), 4096, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
    } else if ((e = hipMemset(cmd, 0, 4096)) != hipSuccess) {
      (void)hipFree(cmd);
      return fail(c, ICPGPU_ERR_HIP, "hipMemset: %s", hipGetErrorString(e));
    } else {
      c->gicp_cmd = cmd;
    }
  }
  c->gicp_resources_ready = true;
  return ICPGPU_OK;
}

// the quadratic inner solver's buffers, when a context first uses it (all or nothing: a failure releases what it had got)
int ensure_gicp_quadratic_resources(icpgpu_ctx* c) {
  if (c->h_quad) return ICPGPU_OK;
  hipError_t e;
  const size_t part_bytes = (size_t)kGicpQuadBlocks * kGicpQuadSums * 2 * sizeof(double);
  const size_t out_bytes = (size_t)(2 * kGicpQuadSums + 8) * 16;  // (+ the development flavour's stamps)
  void* h = nullptr;
  double* partials = nullptr;
  unsigned int* done = nullptr;
  const char* what = "allocation";
  if ((e = hipMalloc(reinterpret_cast<void**>(&partials), part_bytes)) == hipSuccess &&
      (e = hipMalloc(reinterpret_cast<void**>(&done), 64)) == hipSuccess && (e = hipMemset(done, 0, 64)) == hipSuccess &&
      (e = hipHostMalloc(&h, out_bytes, hipHostMallocMapped | hipHostMallocCoherent)) == hipSuccess) {
    std::memset(h, 0, out_bytes);
    what = "hipHostGetDevicePointer";
    e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_quad_dev), h, 0);
  }
  if (e != hipSuccess) {
    if (partials) (void)hipFree(partials);
    if (done) (void)hipFree(done);
    if (h) (void)hipHostFree(h);
    c->h_quad_dev = nullptr;
    return fail(c, e == hipErrorOutOfMemory ? ICPGPU_ERR_OOM : ICPGPU_ERR_HIP, "quadratic GICP solver buffers (%s): %s", what, hipGetErrorString(e));
  }
  c->quad_partials = partials;
  c->quad_done = done;
  c->h_quad = static_cast<volatile unsigned long long*>(h);
  return ICPGPU_OK;
}

bool gicp_inner_quadratic(const icpgpu_ctx* c) {
  static const int forced = [] {  // ICPGPU_GICP_INNER=exact|quadratic overrides the parameter
    const char* e = std::getenv("ICPGPU_GICP_INNER");
    if (!e || !*e) return -1;
    return (e[0] == 'q' || e[0] == 'Q' || e[0] == '1') ? 1 : 0;
  }();
  return forced >= 0 ? forced == 1 : c->params.gicp_inner == ICPGPU_GICP_INNER_QUADRATIC;
}

}  // namespace icpgpu_impl

extern "C" {

int icpgpu_version(void) { return ICPGPU_VERSION_MAJOR * 1000 + ICPGPU_VERSION_MINOR; }

// the library's own defaults
static void default_params_full(icpgpu_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->method = ICPGPU_P2P_SVD;
  p->max_iterations = 10;                    // icp_odometer.h:65
  p->transformation_epsilon = 1e-6;          // icp_odometer.h:64
  p->max_correspondence_distance = 1.0;      // icp_odometer.h:63
  p->euclidean_fitness_epsilon = -DBL_MAX;   // PCL default
  p->min_correspondences = 3;                // PCL default
  p->force_iterations = 0;
  p->nn_mode = ICPGPU_NN_AUTO;
  p->brute_variant = 0;
  p->gicp_inner = ICPGPU_GICP_INNER_EXACT;
}

// ABI rule of include/icpgpu.h: the caller's struct may be shorter (an older header) or longer (a newer one) than the library's
void icpgpu_default_params_sz(icpgpu_params* p, size_t sizeof_params) {
  if (!p || sizeof_params == 0) return;
  icpgpu_params full;
  default_params_full(&full);
  std::memset(p, 0, sizeof_params);
  std::memcpy(p, &full, std::min(sizeof_params, sizeof(full)));
}

void icpgpu_struct_sizes(size_t out3[3]) {
  if (!out3) return;
  out3[0] = sizeof(icpgpu_params);
  out3[1] = sizeof(icpgpu_result);
  out3[2] = sizeof(icpgpu_profile);
}

int icpgpu_create_abi(icpgpu_ctx** out_ctx, int device_id, int header_version, size_t sizeof_params, size_t sizeof_result,
                      size_t sizeof_profile) {
  if (out_ctx) *out_ctx = nullptr;
  if (header_version / 1000 != ICPGPU_VERSION_MAJOR)
    return fail(nullptr, ICPGPU_ERR_UNSUPPORTED, "the caller was built against icpgpu.h %d.%d, this library is %d.%d: another major version",
                header_version / 1000, header_version % 1000, ICPGPU_VERSION_MAJOR, ICPGPU_VERSION_MINOR);
  // (the floor: icp_ctx.h)
  if (sizeof_params < kAbiParams10 || sizeof_result < kAbiResult10 || sizeof_profile < kAbiProfile10 || sizeof_params > 4096 ||
      sizeof_result > 4096 || sizeof_profile > 65536)
    return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "struct sizes %zu / %zu / %zu are not those of an icpgpu.h 1.x (at least %zu / %zu / %zu)",
                sizeof_params, sizeof_result, sizeof_profile, kAbiParams10, kAbiResult10, kAbiProfile10);
  const int rc = create_context(out_ctx, device_id, /*with_stream=*/true);
  if (rc != ICPGPU_OK) return rc;
  (*out_ctx)->abi_params = sizeof_params;
  (*out_ctx)->abi_result = sizeof_result;
  (*out_ctx)->abi_profile = sizeof_profile;
  return ICPGPU_OK;
}

}  // extern "C"

namespace icpgpu_impl {

// A context's own stream, for contexts created without one (batch workers: below)
int ensure_stream(icpgpu_ctx* c) {
  if (c->stream) return ICPGPU_OK;
  HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  return ICPGPU_OK;
}

// icpgpu_create.  with_stream = false: a batch worker that will run inside a lock-step group on the GROUP's stream
// (icpgpu_batch.cpp) -- hipStreamCreate is 0.6-4 ms on this runtime (scripts/probes/create_probe.cpp; everything else a context
// needs comes to ~0.1 ms), and a batch wants up to 64 workers but only eight streams; a worker that ends up on the round-robin
// or GICP path gets its stream then (ensure_stream).
int create_context(icpgpu_ctx** out_ctx, int device_id, bool with_stream) {
  if (!out_ctx) return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "out_ctx is null");
  *out_ctx = nullptr;
  // The device's identity is looked up once per process and device: hipGetDeviceProperties alone is most of a millisecond, and
  // icpgpu_align_batch creates up to 64 worker contexts on its first call.
  struct DeviceInfo {
    int state = 0;  // 0 unknown, 1 a gfx950 device, 2 looked up and unusable
    int cus = 256;
    char arch[64] = {0};
  };
  static std::mutex info_mutex;
  static int info_count = -1;
  static DeviceInfo info[64];
  hipError_t e = hipSuccess;
  int cus = 256;
  {
    std::lock_guard<std::mutex> lock(info_mutex);
    if (info_count < 0) {
      int count = 0;
      e = hipGetDeviceCount(&count);
      if (e != hipSuccess || count <= 0)
        return fail(nullptr, ICPGPU_ERR_NO_DEVICE, "no HIP device available (%s); libicpgpu has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
      info_count = count;
    }
    if (device_id < 0 || device_id >= info_count)
      return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "device_id %d out of range [0, %d)", device_id, info_count);
    DeviceInfo local;
    DeviceInfo& di = device_id < 64 ? info[device_id] : local;
    if (di.state == 0) {
      hipDeviceProp_t prop;
      e = hipGetDeviceProperties(&prop, device_id);
      if (e != hipSuccess) return fail(nullptr, ICPGPU_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
      std::strncpy(di.arch, prop.gcnArchName, sizeof(di.arch) - 1);
      di.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      di.state = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 2;
    }
    if (di.state != 1)
      return fail(nullptr, ICPGPU_ERR_NO_DEVICE, "device %d is %s; libicpgpu is built for gfx950 (MI355X) only", device_id, di.arch);
    cus = di.cus;
  }

  icpgpu_ctx* c = new (std::nothrow) icpgpu_ctx();
  if (!c) return fail(nullptr, ICPGPU_ERR_OOM, "out of host memory");
  c->device = device_id;
  c->num_cus = cus;
  default_params_full(&c->params);
  c->gicp_choice = gicp_device_solver_mode() == 1 ? 2 : 1;  // auto: the host loop for single alignments (include/icpgpu.h, ICPGPU_GICP_DEVICE)
  if (const char* v = ICPGPU_DEV_ENV("ICPGPU_NN_VARIANT")) c->nn_variant = std::atoi(v);

  auto bail = [&](const char* what, hipError_t err) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(err);
    icpgpu_destroy(c);
    return fail(nullptr, err == hipErrorOutOfMemory ? ICPGPU_ERR_OOM : ICPGPU_ERR_HIP, "%s", msg.c_str());
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
  if (with_stream && (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  for (auto& ev : c->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
  // mailbox: 24 doubles the host keeps the current sums (and a few spare slots) in, then the 17 {sum, number} pairs the
  // device writes (reduce_final_kernel)
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_sums), (24 + 2 * kReduceTerms) * sizeof(double),
                         hipHostMallocMapped | hipHostMallocCoherent)) !=  // fine-grained: the polled flags must become visible without a sync
      hipSuccess)
    return bail("hipHostMalloc", e);
  std::memset(c->h_sums, 0, (24 + 2 * kReduceTerms) * sizeof(double));
  if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_sums_dev), c->h_sums, 0)) != hipSuccess)
    return bail("hipHostGetDevicePointer", e);
  c->h_flags = reinterpret_cast<volatile unsigned long long*>(c->h_sums + 24);
  c->h_flags_dev = reinterpret_cast<unsigned long long*>(c->h_sums_dev + 24);
  {
    // How result pairs leave the device (icp_kernels.h): ICPGPU_MAILBOX=pairs|release decides; otherwise the self-test does,
    // once per device and process: 4096 pairs into one slot, read back concurrently; one torn pair selects the release form.
    static std::atomic<int> verdict[64];  // 0 unknown, 1 pairs, 2 release
    const int slot = device_id >= 0 && device_id < 64 ? device_id : 0;
    int v = verdict[slot].load();
    if (const char* m = std::getenv("ICPGPU_MAILBOX")) v = std::strcmp(m, "release") == 0 ? 2 : (std::strcmp(m, "pairs") == 0 ? 1 : v);
    if (v == 0 && !c->stream && (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if (v == 0) {
      volatile unsigned long long* pair = reinterpret_cast<volatile unsigned long long*>(c->h_sums + 22);
      unsigned long long* pair_dev = reinterpret_cast<unsigned long long*>(c->h_sums_dev + 22);
      const int rounds = 4096;
      pair[0] = pair[1] = 0;
      if ((e = launch_mailbox_selftest(pair_dev, rounds, c->stream)) != hipSuccess) return bail("mailbox self-test", e);
      unsigned long long torn = 0, seen = 0, last = 0;
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 1;; ++spins) {
        const unsigned long long tag = pair[1], bits = pair[0], tag2 = pair[1];
        const unsigned long long n = tag >> 24;
        if (tag == tag2 && n >= 1 && n <= (unsigned long long)rounds) {
          if (tag != mailbox_tag(n, bits)) ++torn;
          if (n != last) {
            ++seen;
            last = n;
          }
        }
        if (n == (unsigned long long)rounds) break;
        if ((spins & 0xFFFu) == 0 &&
            (hipStreamQuery(c->stream) != hipErrorNotReady ||
             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 2000.0))
          break;
      }
      if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("mailbox self-test", e);
      v = torn ? 2 : 1;
      if (std::getenv("ICPGPU_DEBUG"))
        fprintf(stderr, "[icpgpu] mailbox self-test on device %d: %llu of %d pairs observed, %llu torn -> %s\n", device_id, seen, rounds, torn,
                v == 2 ? "release form" : "16-byte pairs");
      pair[0] = pair[1] = 0;
      verdict[slot].store(v);
    }
    c->mailbox_release = v == 2;
  }
  c->ev_ring.assign((size_t)kEventRing * 3, nullptr);  // (created when a slot is first timed: sweep_issue)
  c->pending.reserve(kEventRing);
  {
    void* hp = nullptr;
    // 32 result pairs for fetch_ints and the markers, 8 more for the statistics of a covariance grid built ahead of them (spec_grid)
    // ... and 8 for the marker of a cloud staged for the host (stage_post)
    if ((e = hipHostMalloc(&hp, 48 * 16, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc", e);
    std::memset(hp, 0, 48 * 16);
    c->h_post = static_cast<volatile unsigned long long*>(hp);
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_post_dev), hp, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", e);
  }
  if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_ints), 16 * sizeof(int), hipHostMallocDefault)) != hipSuccess)
    return bail("hipHostMalloc", e);
  if ((e = hipMalloc(&c->partials.ptr, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double))) != hipSuccess)
    return bail("hipMalloc(partials)", e);
  c->partials.cap = (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double);
  if ((e = hipMalloc(&c->sums.ptr, kReduceTerms * sizeof(double))) != hipSuccess) return bail("hipMalloc(sums)", e);
  c->sums.cap = kReduceTerms * sizeof(double);
  *out_ctx = c;
  return ICPGPU_OK;
}

}  // namespace icpgpu_impl

extern "C" {

int icpgpu_destroy(icpgpu_ctx* c) {
  if (c && c->pt_n)
    fprintf(stderr, "[icpgpu] point-to-point sweeps waited for: %llu; per sweep: waiting for the sums %.2f us, sums -> sweep_issue (take, solve, "
                    "convergence) %.2f us, sweep_issue up to the search launch %.2f us, the launch call %.2f us, the rest of sweep_issue "
                    "(final-reduction launch, events) %.2f us\n",
            c->pt_n, c->pt_wait / c->pt_n, c->pt_solve / c->pt_n, c->pt_prelaunch / c->pt_n, c->pt_launch / c->pt_n, c->pt_rest / c->pt_n);
  if (c && c->gt_n)
    fprintf(stderr, "[icpgpu] GICP evaluations through the server: %llu; per evaluation: command write %.2f us, wait for the flags %.2f us "
                    "(first workgroup's result visible -> all visible and checked: %.2f us), merge %.2f us, solver between evaluations %.2f us | "
                    "device, mean over the workgroups: polling %.2f us, accumulate %.2f us, reduce + store %.2f us (three evaluations are "
                    "printed workgroup by workgroup above)\n",
            c->gt_n, c->gt_cmd / c->gt_n, c->gt_wait / c->gt_n, c->gt_trickle / c->gt_n, c->gt_merge / c->gt_n, c->gt_between / c->gt_n,
            c->gt_dev_n ? c->gt_dev_wait / c->gt_dev_n : 0.0, c->gt_dev_n ? c->gt_dev_work / c->gt_dev_n : 0.0,
            c->gt_dev_n ? c->gt_dev_reduce / c->gt_dev_n : 0.0);
  if (c && c->gt_aligns)
    fprintf(stderr, "[icpgpu] GICP alignments: %llu; host wall per alignment (us): covariances %.1f | grid + buffers %.1f | search + Mahalanobis "
                    "launches %.1f | server start + first evaluation %.1f | BFGS %.1f | server stop %.1f | event read-back %.1f | fitness %.1f | "
                    "output cloud %.1f\n",
            c->gt_aligns, c->gt_stage[0] / c->gt_aligns, c->gt_stage[1] / c->gt_aligns, c->gt_stage[2] / c->gt_aligns, c->gt_stage[3] / c->gt_aligns,
            c->gt_stage[4] / c->gt_aligns, c->gt_stage[5] / c->gt_aligns, c->gt_stage[6] / c->gt_aligns, c->gt_stage[7] / c->gt_aligns,
            c->gt_stage[8] / c->gt_aligns);
  if (!c) return ICPGPU_OK;
  for (icpgpu_ctx* w : c->workers) icpgpu_destroy(w);
  c->workers.clear();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  release(c->src.buf);
  release(c->tgt.buf);
  release(c->keys);
  release(c->partials);
  release(c->sums);
  release(c->out);
  for (GridIndex* G : {&c->cov_grid_src, &c->cov_grid_tgt}) {
    release(G->sorted);
    release(G->cell_start);
    release(G->cell_of_point);
    release(G->rank);
    release(G->block_sums);
    release(G->ints);
    release(G->unmatched);
    release(G->leftover);
  }
  release(c->cov_src);
  release(c->cov_tgt);
  release(c->maha);
  release(c->cov_list);
  release(c->vox_in);
  release(c->vox_out);
  release(c->vox_keys);
  release(c->vox_vals);
  release(c->vox_flags);
  release(c->vox_slots);
  release(c->vox_temp);
  release(c->vox_ints);
  release(c->vox_bins);
  release(c->vox_pub);
  release(c->idx);
  release(c->d2);
  release(c->brute_seed.keys);
  release(c->brute_order.pts);
  release(c->brute_order.work);
  release(c->brute_order.check);
  release(c->tile_seed.keys);
  release(c->tile_seed.stats);
  release(c->tile_seed.prev);
  release(c->fp_acc);
  release(c->batch_table);
  release(c->map.node_keys);
  release(c->map.node_vals);
  if (c->cand_counter.ptr) grid_count_candidates(nullptr);
  release(c->cand_counter);
  for (DeviceBuf* b : {&c->map.pts.buf, &c->map.keys, &c->map.vals, &c->map.first, &c->map.staged, &c->map.moved, &c->map.slot_of,
                       &c->map.flags, &c->map.rank, &c->map.temp, &c->map.counter, &c->map.nn_keys, &c->map.first_user,
                       &c->map.uflags, &c->map.urank, &c->map.uniq_index, &c->map.uniq.buf})
    release(*b);
  for (GridIndex* G : {&c->grid, &c->src_grid, &c->map.grid}) {
    release(G->sorted);
    release(G->cell_start);
    release(G->cell_of_point);
    release(G->rank);
    release(G->block_sums);
    release(G->ints);
    release(G->unmatched);
    release(G->leftover);
  }
  if (c->h_sums) (void)hipHostFree(c->h_sums);
  if (c->h_gicp) (void)hipHostFree(c->h_gicp);
  if (c->gicp_cmd) (void)hipFree(c->gicp_cmd);
  if (c->gicp_slots) (void)hipFree(c->gicp_slots);
  if (c->gicp_slots_local) (void)hipFree(c->gicp_slots_local);
  if (c->gicp_owner) (void)hipFree(c->gicp_owner);
  if (c->h_solve) (void)hipHostFree(const_cast<unsigned long long*>(c->h_solve));
  if (c->h_quad) (void)hipHostFree(const_cast<unsigned long long*>(c->h_quad));
  if (c->quad_partials) (void)hipFree(c->quad_partials);
  if (c->quad_done) (void)hipFree(c->quad_done);
  if (c->h_ints) (void)hipHostFree(c->h_ints);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_post) (void)hipHostFree(const_cast<unsigned long long*>(c->h_post));
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_ring)
    if (ev) (void)hipEventDestroy(ev);
  for (hipStream_t st : c->group_streams) (void)hipStreamDestroy(st);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ICPGPU_OK;
}

const char* icpgpu_last_error(const icpgpu_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int icpgpu_set_params(icpgpu_ctx* c, const icpgpu_params* user) {
  if (!c || !user) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  icpgpu_params full;  // fields the caller's (older) header does not have keep their defaults; fields of a newer header are ignored
  default_params_full(&full);
  std::memcpy(&full, user, std::min(c->abi_params, sizeof(full)));
  const icpgpu_params* p = &full;
  if (p->method != ICPGPU_P2P_SVD && p->method != ICPGPU_GICP) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad method");
  if (p->nn_mode < ICPGPU_NN_AUTO || p->nn_mode > ICPGPU_NN_GRID) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad nn_mode");
  if (p->brute_variant < 0 || p->brute_variant > 2) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad brute_variant");
  if (p->gicp_inner != ICPGPU_GICP_INNER_EXACT && p->gicp_inner != ICPGPU_GICP_INNER_QUADRATIC)
    return fail(c, ICPGPU_ERR_INVALID_ARG, "bad gicp_inner");
  c->params = *p;
  return ICPGPU_OK;
}

int icpgpu_get_params(const icpgpu_ctx* c, icpgpu_params* p) {
  if (!c || !p) return ICPGPU_ERR_INVALID_ARG;
  std::memset(p, 0, c->abi_params);
  std::memcpy(p, &c->params, std::min(c->abi_params, sizeof(c->params)));
  return ICPGPU_OK;
}


// setInputSource from a host buffer.  The reference filters every scan first and hands the filtered cloud over as the source
// (`voxelFilterCloud(&curr_cloud_, ...)`, icp_odometer.cpp:177, then setInputSource at :193): when this context ran that filter
// (icpgpu_voxel_grid / icpgpu::VoxelGrid::filter) the very bytes are still in HBM.  A buffer of the filtered cloud's size whose
// content fingerprint equals the one taken when it was fetched is adopted from there -- a device-to-device copy queued on the
// stream instead of an upload and a stream synchronisation -- together with the raw scan's bounding box, so that the grids
// built over it need no bounding-box pass of their own: the sequence then costs what icpgpu_set_source_voxel_filtered costs.
// Same bits in HBM either way; same ASSUMPTION and the same switch (ICPGPU_RECOGNISE=0) as icpgpu_set_target's recognition.
int icpgpu_set_source(icpgpu_ctx* c, const float* xyzw, size_t n) {
  ENTER(c);
  if (recognise_enabled() && c->vox_fp_valid && n > 0 && xyzw && n == c->vox_last_n && c->vox_out.ptr &&
      sample_fingerprint(xyzw, n) == c->vox_sample_fp && icpgpu_fingerprint(xyzw, n) == c->vox_fp) {
    if (c->src.buf.external) c->src.buf = DeviceBuf{};
    int rc = ensure(c, c->src.buf, n * sizeof(float4));
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->src.buf.ptr, c->vox_out.ptr, n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    c->src_version++;
    c->src.n = n;
    c->src.set = true;
    c->src.sample_fp = c->vox_sample_fp;
    c->src.sample_valid = true;
    c->src.bbox_version = 0;
    if (c->vox_box_valid) {  // a box that CONTAINS the centroids (icpgpu_set_source_voxel_filtered)
      std::memcpy(c->src.bbox_enc, c->vox_box, sizeof(c->vox_box));
      c->src.bbox_version = c->src_version;
      c->src.bbox_exact = false;
    }
    c->src_fp = c->vox_fp;
    c->src_fp_version = c->src_version;
    c->prof.sources_adopted += 1;
    return ICPGPU_OK;
  }
  c->src_version++;
  return set_cloud_host(c, c->src, xyzw, n);
}

// setInputTarget from a host buffer.  The reference's odometer hands over, as the target of scan k, the cloud it handed
// over as the source of scan k - 1 (`*prev_cloud_ = *curr_cloud_`, icp_odometer.cpp:209, then :194 on the next scan) -- a
// fresh registration object cannot know that, the context can: when the buffer has the size of a cloud it already holds
// it compares content fingerprints (one pass over the host buffer, ~0.1 ms per MB, + one 8-byte read-back) and
//   * keeps the current target with its grid and GICP covariances when the content is the same (a rejected scan keeps
//     prev_cloud_: icp_odometer.cpp:201-210), or
//   * takes the promote path when it is the current SOURCE's content (no upload, grid and covariances carried over).
// Results are identical either way: the target cloud in HBM holds the same bits.  External (zero-copy) buffers never take
// part.  Callers that replace the source in the same step must set the target FIRST (the C++ shim does).
int icpgpu_set_target(icpgpu_ctx* c, const float* xyzw, size_t n) {
  ENTER(c);
  if (recognise_enabled() && n > 0 && xyzw) {
    bool tgt_cand = c->tgt.set && c->tgt.n == n && !c->tgt.buf.external;
    bool src_cand = c->src.set && c->src.n == n && !c->src.buf.external;
    if ((tgt_cand && c->tgt.sample_valid) || (src_cand && c->src.sample_valid)) {  // a microsecond's look before the real one
      const unsigned long long sf = sample_fingerprint(xyzw, n);
      if (tgt_cand && c->tgt.sample_valid && c->tgt.sample_fp != sf) tgt_cand = false;
      if (src_cand && c->src.sample_valid && c->src.sample_fp != sf) src_cand = false;
    }
    if (tgt_cand || src_cand) {
      const unsigned long long fp = icpgpu_fingerprint(xyzw, n);
      unsigned long long have = 0;
      int rc;
      if (tgt_cand) {
        if ((rc = device_fingerprint(c, c->tgt, c->tgt_version, c->tgt_fp, c->tgt_fp_version, &have))) return rc;
        if (have == fp) {
          c->prof.targets_recognised += 1;
          return ICPGPU_OK;
        }
      }
      if (src_cand) {
        if ((rc = device_fingerprint(c, c->src, c->src_version, c->src_fp, c->src_fp_version, &have))) return rc;
        if (have == fp) {
          c->prof.targets_recognised += 1;
          rc = promote_internal(c);
          if (rc != ICPGPU_OK) return rc;
          c->tgt_fp = fp;
          c->tgt_fp_version = c->tgt_version;
          // set_target must not take the source away (a caller may set the SAME cloud as source and target, or set the source
          // first): the source is put back as a device-to-device copy of what is now the target -- microseconds, and the
          // odometer's next set_source overwrites it anyway.  Its cell order / covariances moved on with the target.
          if ((rc = ensure(c, c->src.buf, n * sizeof(float4)))) return rc;
          HIP_TRY(c, hipMemcpyAsync(c->src.buf.ptr, c->tgt.buf.ptr, n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
          c->src.n = n;
          c->src.set = true;
          c->src.sample_fp = c->tgt.sample_fp;  // the copy IS the target's content
          c->src.sample_valid = c->tgt.sample_valid;
          c->src.bbox_version = 0;
          c->src_fp = fp;
          c->src_fp_version = c->src_version;
          return ICPGPU_OK;
        }
      }
      c->tgt_version++;
      rc = set_cloud_host(c, c->tgt, xyzw, n);
      if (rc == ICPGPU_OK) {  // what was just uploaded has the fingerprint just computed
        c->tgt_fp = fp;
        c->tgt_fp_version = c->tgt_version;
      }
      return rc;
    }
  }
  c->tgt_version++;
  return set_cloud_host(c, c->tgt, xyzw, n);
}
int icpgpu_set_source_device(icpgpu_ctx* c, const void* d, size_t n) {
  ENTER(c);
  c->src_version++;
  return set_cloud_device(c, c->src, d, n);
}
int icpgpu_set_target_device(icpgpu_ctx* c, const void* d, size_t n) {
  ENTER(c);
  c->tgt_version++;
  return set_cloud_device(c, c->tgt, d, n);
}

int icpgpu_promote_source_to_target(icpgpu_ctx* c) {
  ENTER(c);
  return promote_internal(c);
}

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)  // (host code: this unit goes through hipcc's device pass as well)
#define ICPGPU_FP_AVX512 1
// The same sum, eight 64-bit words (four points) per step: lane l holds word l % 2 of point i + l / 2, the even lanes take the
// w0 branch of fp_point, the odd ones the w1 branch; the per-lane multiples of the two odd constants advance by additions.
// ~4x the scalar loop's 5 GB/s (one pass over a 200k-point scan: 0.64 -> 0.17 ms) where the CPU has AVX-512 DQ (vpmullq).
__attribute__((target("avx512f,avx512dq"))) static unsigned long long fingerprint_sum_avx512(const unsigned char* b, size_t n, size_t* done) {
  const unsigned long long K1 = 0x9e3779b97f4a7c15ull, K2 = 0xd6e8feb86659fd93ull;
  // lane l: A = K1 * (2 i + 1) (even) or K2 * (2 i + 2) (odd) for i = l / 2; one step = four points further
  alignas(64) unsigned long long a0[8], step[8];
  for (int l = 0; l < 8; ++l) {
    const unsigned long long i = (unsigned long long)(l / 2);
    a0[l] = (l & 1) ? K2 * (2ull * i + 2ull) : K1 * (2ull * i + 1ull);
    step[l] = (l & 1) ? K2 * 8ull : K1 * 8ull;
  }
  __m512i A = _mm512_load_si512(a0), acc0 = _mm512_setzero_si512(), acc1 = _mm512_setzero_si512();
  const __m512i S = _mm512_load_si512(step), S2 = _mm512_add_epi64(S, S);
  const __m512i M1 = _mm512_set1_epi64((long long)0xbf58476d1ce4e5b9ull), M2 = _mm512_set1_epi64((long long)0x94d049bb133111ebull);
  const __mmask8 odd = 0xAA;
  auto mix = [&](__m512i x) __attribute__((target("avx512f,avx512dq"))) {
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 30));
    x = _mm512_mullo_epi64(x, M1);
    x = _mm512_xor_si512(x, _mm512_srli_epi64(x, 27));
    x = _mm512_mullo_epi64(x, M2);
    return _mm512_xor_si512(x, _mm512_srli_epi64(x, 31));
  };
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {  // two independent steps per trip
    const __m512i w0 = _mm512_loadu_si512(b + 16 * i), w1 = _mm512_loadu_si512(b + 16 * i + 64);
    const __m512i B = _mm512_add_epi64(A, S);
    const __m512i x0 = _mm512_mask_xor_epi64(_mm512_add_epi64(w0, A), odd, w0, A);
    const __m512i x1 = _mm512_mask_xor_epi64(_mm512_add_epi64(w1, B), odd, w1, B);
    acc0 = _mm512_add_epi64(acc0, mix(x0));
    acc1 = _mm512_add_epi64(acc1, mix(x1));
    A = _mm512_add_epi64(A, S2);
  }
  *done = i;
  return (unsigned long long)_mm512_reduce_add_epi64(_mm512_add_epi64(acc0, acc1));
}
static bool fingerprint_has_avx512() {
  static const bool v = [] {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
  }();
  return v;
}
#endif

unsigned long long icpgpu_fingerprint(const float* xyzw, size_t n) {
  unsigned long long s0 = 0, s1 = 0;
  const unsigned char* b = reinterpret_cast<const unsigned char*>(xyzw);
  size_t i = 0;
#if defined(ICPGPU_FP_AVX512)
  if (n >= 64 && fingerprint_has_avx512()) s0 = fingerprint_sum_avx512(b, n, &i);
#endif
  for (; i < n; ++i) {  // (two independent multiply chains per point: ~5 GB/s on one core)
    unsigned long long w0, w1;
    std::memcpy(&w0, b + 16 * i, 8);
    std::memcpy(&w1, b + 16 * i + 8, 8);
    s0 += fp_mix(w0 + 0x9e3779b97f4a7c15ull * (2ull * i + 1ull));
    s1 += fp_mix(w1 ^ (0xd6e8feb86659fd93ull * (2ull * i + 2ull)));
  }
  return fp_finish(s0 + s1, (unsigned long long)n);
}

int icpgpu_cloud_sizes(const icpgpu_ctx* c, size_t* n_source, size_t* n_target) {
  if (!c) return ICPGPU_ERR_INVALID_ARG;
  if (n_source) *n_source = c->src.set ? c->src.n : 0;
  if (n_target) *n_target = c->tgt.set ? c->tgt.n : 0;
  return ICPGPU_OK;
}

}  // extern "C"
namespace icpgpu_impl {
int promote_internal(icpgpu_ctx* c) {
  if (!c->src.set) return fail(c, ICPGPU_ERR_NO_INPUT, "promote_source_to_target: no source set");
  std::swap(c->src, c->tgt);
  c->tgt_fp = c->src_fp;  // (versions are re-stamped below)
  const bool fp_follows = c->src_fp_version == c->src_version;
  // the source's GICP covariances stay valid for the cloud that is now the target
  std::swap(c->cov_src, c->cov_tgt);
  std::swap(c->cov_grid_src, c->cov_grid_tgt);
  const bool cov_grid_follows = c->cov_grid_tgt.built && c->cov_grid_tgt.version == c->src_version;
  const bool box_follows = c->tgt.bbox_version != 0 && c->tgt.bbox_version == c->src_version;  // (c->tgt is the old source by now)
  // ... and so does its cell order: it is the new target's grid
  std::swap(c->grid, c->src_grid);
  const bool grid_follows = c->grid.built && c->grid.version == c->src_version;
  c->tgt_version++;
  c->tgt_fp_version = fp_follows ? c->tgt_version : 0;
  c->src_fp_version = 0;
  if (grid_follows) c->grid.version = c->tgt_version;
  else c->grid.built = c->grid.usable = false;
  c->src_grid.built = c->src_grid.usable = false;
  c->cov_tgt_version = (c->cov_src_version == c->src_version) ? c->tgt_version : 0;
  if (cov_grid_follows) c->cov_grid_tgt.version = c->tgt_version;  // (ensure_grid may adopt it for the GICP search)
  else c->cov_grid_tgt.built = c->cov_grid_tgt.usable = false;
  c->cov_grid_src.built = c->cov_grid_src.usable = false;          // the old target's: never to be mistaken for a new source's
  c->tgt.bbox_version = box_follows ? c->tgt_version : 0;          // the box travels with the cloud, re-stamped for its new counter
  c->src.bbox_version = 0;
  c->cov_src_version = 0;
  c->src_version++;
  c->src.n = 0;
  c->src.set = false;
  c->src.sample_valid = false;  // (the swap left the old target's sample here)
  if (c->src.buf.external) c->src.buf = DeviceBuf{};
  c->have_final = false;
  return ICPGPU_OK;
}
}  // namespace icpgpu_impl
extern "C" {

int icpgpu_profile_reset(icpgpu_ctx* c) {
  if (!c) return ICPGPU_ERR_INVALID_ARG;
  (void)resolve_sweep_timings(c);
  (void)resolve_cov_timing(c);
  std::memset(&c->prof, 0, sizeof(c->prof));
  for (double& v : c->gt_stage) v = 0.0;  // (ICPGPU_GICP_TIMING's host stage timers restart too: a harness resets after its warm-up)
  c->gt_aligns = 0;
  return ICPGPU_OK;
}

int icpgpu_profile_set_sampling(icpgpu_ctx* c, int every) {
  if (!c || every < 1) return ICPGPU_ERR_INVALID_ARG;
  c->timing_every = every;
  c->sweep_counter = 0;
  for (icpgpu_ctx* w : c->workers) {
    w->timing_every = every;
    w->sweep_counter = 0;
  }
  return ICPGPU_OK;
}

int icpgpu_profile_get(icpgpu_ctx* c, icpgpu_profile* out) {
  if (!c || !out) return ICPGPU_ERR_INVALID_ARG;
  int rc = resolve_sweep_timings(c);
  if (rc) return rc;
  if ((rc = resolve_cov_timing(c))) return rc;
  // which inner solver this context's single GICP alignments run on (icpgpu_gicp.cpp): fixed at creation, changed by icpgpu_calibrate
  // only; a device solver that gave up on this context (a gather timed out) reads as the host loop
  c->prof.gicp_solver_choice = (c->gicp_choice == 2 && c->gicp_resources_ready && !c->gicp_device_ok) ? 1u : (uint64_t)c->gicp_choice;
  std::memset(out, 0, c->abi_profile);
  std::memcpy(out, &c->prof, std::min(c->abi_profile, sizeof(c->prof)));
  return ICPGPU_OK;
}

int icpgpu_get_stream(icpgpu_ctx* c, void** out_stream) {
  if (!c || !out_stream) return ICPGPU_ERR_INVALID_ARG;
  *out_stream = static_cast<void*>(c->stream);
  return ICPGPU_OK;
}

int icpgpu_synchronize(icpgpu_ctx* c) {
  ENTER(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ICPGPU_OK;
}

int icpgpu_count_candidates(icpgpu_ctx* c, int enable) {
  ENTER(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (enable) {
    int rc = ensure(c, c->cand_counter, sizeof(unsigned long long));
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->cand_counter.ptr, 0, sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    grid_count_candidates(static_cast<unsigned long long*>(c->cand_counter.ptr));
  } else {
    grid_count_candidates(nullptr);
  }
  return ICPGPU_OK;
}

int icpgpu_count_candidates_read(icpgpu_ctx* c, uint64_t* out) {
  ENTER(c);
  if (!out) return fail(c, ICPGPU_ERR_INVALID_ARG, "out is null");
  if (!c->cand_counter.ptr) return fail(c, ICPGPU_ERR_NO_INPUT, "count_candidates_read: counting was never enabled on this context");
  unsigned long long v = 0;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(&v, c->cand_counter.ptr, sizeof(v), hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemset(c->cand_counter.ptr, 0, sizeof(v)));
  *out = (uint64_t)v;
  return ICPGPU_OK;
}


}  // extern "C"
