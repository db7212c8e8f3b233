// icpgpu_gicp.cpp -- GICP mode behind icpgpu_align (pcl::GeneralizedIterativeClosestPoint, what the reference instantiates at
// icp_odometer.cpp:188 and octree_mapper.cpp:104; kernels: icp_gicp.hip, solver: icp_gicp_solver.cpp).
#include "icp_ctx.h"


namespace icpgpu_impl {

// ---- GICP mode: pcl::GeneralizedIterativeClosestPoint::computeTransformation ----------------------
void mat4f_identity(float m[16]) {
  for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}
void mat4f_mul(const float a[16], const float b[16], float out[16]) {  // column-major, float accumulate (Eigen Matrix4f)
  float r[16];
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + row] * b[col * 4 + k];
      r[col * 4 + row] = s;
    }
  std::memcpy(out, r, sizeof(r));
}
Xform xform_from_f16(const float f[16]) {
  Xform x;
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 4; ++cc) x.m[4 * r + cc] = f[cc * 4 + r];
  return x;
}

// per-point covariances of `cloud` (20-NN in its own grid); cached per cloud version.  In two halves, so that a resumable run
// (GicpRun, below) can take the grid build through its host round trips without blocking: cov_grid_begin starts the build of the
// cloud's k-NN grid (needed = false: the covariances are cached, nothing to do), cov_launch -- once the build is Done -- queues
// the covariance pass.  ensure_covariances is the two with the blocking build loop in between.
static int cov_grid_begin(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, GridIndex& G, const DeviceBuf& cov, uint64_t cov_version,
                          GridBuild& b, bool& needed) {
  needed = !(cov_version == version && cov.ptr);
  if (!needed) return ICPGPU_OK;
  // the covariance search has no distance cap; the grid only needs cells of a useful size: same rule as the NN grid
  const double cut = std::max(1e-3, c->params.max_correspondence_distance);
  // point-weighted cell population the covariance grid aims at.  8 until the kernel walked a cube level as one dense list of
  // candidates (round 4); re-measured then: a voxel-filtered 22k-point scan 110 (8) / 94 (16) / 96 (24) / 98 us (32), raw scans --
  // whose density falls with the square of the range, so that cells sized for the near field are empty in the far field --
  // 5k 0.95 / 0.58 / 0.52, 50k 0.53 / 0.48 / 0.31, 200k 1.21 / 1.14 / 0.92 ms per cloud at 8 / 16 / 32.  (The neighbours are exact
  // whatever the cells: a tuning constant, not a result.)
  static const double knn_pop = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_KNN_POP"); return e ? std::atof(e) : 32.0; }();
  // The first count pass starts from the cell size the LAST covariance grid of this context settled on (consecutive scans of a
  // drive are alike: one pass instead of two); the rule that accepts or corrects it is the same.  History may decide the cells,
  // never a result: the 20 neighbours are exact whatever the cells are, and they are summed in order of distance.
  static const bool hint_enabled = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_KNN_HINT"); return !e || std::atoi(e) != 0; }();
  const double cut_used = std::isfinite(cut) ? std::min(cut, 1e6) : 1.0;
  const double h_start = (hint_enabled && c->cov_h_hint > 0.0 && c->cov_h_hint_cut == cut_used) ? c->cov_h_hint : 0.0;
  return gb_begin(c, b, cloud, version, cut_used, /*adapt=*/false, G, nullptr, knn_pop, h_start);
}

static int cov_launch(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, GridIndex& G, DeviceBuf& cov, uint64_t& cov_version, bool timed) {
  if (G.usable) {
    const double cut = std::max(1e-3, c->params.max_correspondence_distance);
    c->cov_h_hint = (double)G.g.h;
    c->cov_h_hint_cut = std::isfinite(cut) ? std::min(cut, 1e6) : 1.0;
  }
  // a cloud the k-NN grid refuses (thousands of points in one cell of the largest table: tight clusters in a wide volume) has its
  // covariances computed without one, every point by the far-field kernel's passes over the whole cloud -- up to kGicpCovFarMost points
  if (!G.usable && cloud.n > (size_t)kGicpCovFarMost)
    return fail(c, ICPGPU_ERR_UNSUPPORTED, "GICP: cannot index this cloud of %zu points (degenerate or non-finite input)", cloud.n);
  int rc;
  if ((rc = ensure(c, cov, cloud.n * 6 * sizeof(double)))) return rc;
  if (timed) {
    if ((rc = resolve_cov_timing(c))) return rc;  // (the events are about to be reused: an alignment with two new clouds)
    HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  }
  if ((rc = ensure(c, c->cov_list, (2 * cloud.n + 2) * sizeof(int)))) return rc;
  const bool counters_zero = c->cov_list_zeroed == c->cov_list.ptr && c->cov_list_zeroed_cap == c->cov_list.cap;
  c->cov_list_zeroed = nullptr;  // (until the launches below are known to be queued)
  if (G.usable)
    HIP_TRY(c, launch_gicp_covariances(cloud.data(), (int)cloud.n, static_cast<const float4*>(G.sorted.ptr),
                                       static_cast<const int*>(G.cell_start.ptr), G.g, static_cast<double*>(cov.ptr), c->stream,
                                       static_cast<int*>(c->cov_list.ptr), counters_zero));
  else
    HIP_TRY(c, launch_gicp_covariances_brute(cloud.data(), (int)cloud.n, static_cast<double*>(cov.ptr), c->stream,
                                             static_cast<int*>(c->cov_list.ptr), counters_zero));
  c->cov_list_zeroed = c->cov_list.ptr;
  c->cov_list_zeroed_cap = c->cov_list.cap;
  if (timed) {
    HIP_TRY(c, hipEventRecord(c->ev[3], c->stream));
    // No synchronisation: what follows is queued behind the pass (it used to end with one only to time itself: the host sat out
    // the ~0.1 ms of the kernel instead of queueing the search, the Mahalanobis kernel and the evaluation server meanwhile).
    c->cov_timing_pending = true;
  }
  c->prof.gicp_cov_launches += 1;
  c->prof.gicp_cov_points += (uint64_t)cloud.n;
  cov_version = version;
  return ICPGPU_OK;
}

// allow_unchecked (align_gicp, the source cloud): when the cloud brings a containing box (the voxel filter's) and the context a proven
// cell size (the last covariance grid's), the build does not wait for the count pass's statistics -- count, scan, scatter and the
// covariance pass are queued in one go, the statistics are posted behind the count pass and checked by covariance_grid_check() when
// the alignment first waits for the device anyway.  (Per scan of the reference's pipeline this was the one remaining host round trip
// of the index build: ~110 us of host wall in front of a 96 us kernel)
int ensure_covariances(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, GridIndex& G, DeviceBuf& cov, uint64_t& cov_version, bool allow_unchecked) {
  GridBuild b;
  b.post = true;  // (the read-backs go through fetch_ints, as in build_grid)
  bool needed = false;
  int rc = cov_grid_begin(c, cloud, version, G, cov, cov_version, b, needed);
  if (rc || !needed) return rc;
  static const bool unchecked_enabled = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_COV_GRID_UNCHECKED"); return !e || std::atoi(e) != 0; }();
  if (allow_unchecked && unchecked_enabled && b.state == GridBuild::WaitCount && b.h_start > 0.0 && !c->spec_grid.pending) {
    if (c->spec_cooldown > 0) {
      c->spec_cooldown -= 1;
    } else {
      const int* d_ints = static_cast<const int*>(G.ints.ptr);
      const unsigned long long number = ++c->post_seq;
      HIP_TRY(c, launch_post_ints(d_ints + 6, kGridStatInts, c->h_post_dev + 2 * 32, wire_seq(c, number), c->stream));
      if ((rc = gb_finish_unchecked(c, b))) return rc;
      c->spec_grid.pending = true;
      c->spec_grid.number = number;
      c->spec_grid.serial = G.serial;
      c->spec_grid.n = (int)cloud.n;
      c->spec_grid.h = b.h;
      c->spec_grid.knn_population = b.knn_population;
      c->spec_grid.cut = b.cut;
      c->prof.cov_grids_unchecked += 1;
      return cov_launch(c, cloud, version, G, cov, cov_version, /*timed=*/true);
    }
  }
  while (!rc && b.state != GridBuild::Done) {
    const int* d_ints = static_cast<const int*>(G.ints.ptr);
    if (b.state == GridBuild::WaitBbox) rc = fetch_ints(c, d_ints, 6, c->h_ints);
    else rc = fetch_ints(c, d_ints + 6, kGridStatInts, c->h_ints + 6);
    if (!rc) rc = gb_advance(c, b);
  }
  if (rc) return rc;
  return cov_launch(c, cloud, version, G, cov, cov_version, /*timed=*/true);
}

// The statistics of a covariance grid that was built ahead of them (above): 0 = none pending or all is well (the grid's
// population figures and the context's cell-size hint are brought up to date), 1 = the build has to be repeated the waiting way
// (a point outside the box it was given, a non-finite point, a cell beyond kMaxCellPopulation: the source's covariances and its
// grid are invalidated, the caller starts over), < 0 = error.  The post was queued right behind the count pass: by the time an
// alignment has queued its covariance pass, its first search and its first evaluation the pairs have long arrived.
int covariance_grid_check(icpgpu_ctx* c) {
  if (!c->spec_grid.pending) return 0;
  c->spec_grid.pending = false;
  int v[kGridStatInts];
  {
    const int wrc = wait_posted(c, c->h_post + 2 * 32, kGridStatInts, c->spec_grid.number, v, "a grid's statistics");
    if (wrc) return wrc;
  }
  unsigned long long sumsq = 0;
  std::memcpy(&sumsq, v + 2, sizeof sumsq);
  const int max_pop = v[1], binned = v[4];
  GridIndex* G = c->cov_grid_src.serial == c->spec_grid.serial ? &c->cov_grid_src : (c->cov_grid_tgt.serial == c->spec_grid.serial ? &c->cov_grid_tgt : nullptr);
  // development flavour, ICPGPU_COV_GRID_UNCHECKED=2: every check fails (the tests of the start-over path)
  static const bool fail_all = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_COV_GRID_UNCHECKED"); return e && std::atoi(e) == 2; }();
  if (binned != c->spec_grid.n || max_pop > kMaxCellPopulation || fail_all) {
    c->prof.cov_grids_rebuilt += 1;
    c->spec_cooldown = fail_all ? 1 : 64;
    if (std::getenv("ICPGPU_DEBUG"))
      fprintf(stderr, "[icpgpu] covariance grid built ahead of its statistics: %d of %d points binned, largest cell %d -> rebuilt the waiting way\n", binned,
              c->spec_grid.n, max_pop);
    // whichever role the cloud has by now: its grid, its covariances and its cached box are not to be trusted
    for (int role = 0; role < 2; ++role) {
      GridIndex& V = role ? c->cov_grid_tgt : c->cov_grid_src;
      if (V.serial != c->spec_grid.serial) continue;
      V.built = V.usable = false;
      (role ? c->cov_tgt_version : c->cov_src_version) = 0;
      (role ? c->tgt : c->src).bbox_version = 0;
    }
    return 1;
  }
  const double pop = binned > 0 ? (double)sumsq / (double)binned : 0.0;
  if (G) {
    G->max_pop = max_pop;
    G->point_population = pop;
  }
  // the hint follows the rule gb_on_count applies to a first count pass: a population within a factor of two of the aim keeps the
  // cells, another one gets the cells it would have been re-counted with -- for the NEXT cloud
  const double aim = c->spec_grid.knn_population, h = c->spec_grid.h;
  if (aim > 0.0 && pop > 0.0 && (pop > 2.0 * aim || pop < 0.5 * aim))
    c->cov_h_hint = std::min(std::max(h * std::sqrt(aim / pop), c->spec_grid.cut / 64.0), 8.0 * h);
  return 0;
}

// the duration of the last covariance pass into the profile (waits for its end if need be: callers sit behind a result anyway)
int resolve_cov_timing(icpgpu_ctx* c) {
  if (!c->cov_timing_pending) return ICPGPU_OK;
  c->cov_timing_pending = false;
  HIP_TRY(c, hipEventSynchronize(c->ev[3]));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
  c->prof.gicp_cov_ms += ms;
  return ICPGPU_OK;
}

// The host's merge of an evaluation's answers (validated lines of nblk workgroups, icp_kernels.h: gicp_line_value -- value s of a
// block sits at word 8 (s / 7) + s % 7: [0] m, [1..13] the high parts, [14] sum d2, [15..27] the low parts): workgroup by workgroup,
// in double-double like the kernel (icp_gicp.hip) -- (hi, lo) += (bh, bl) is a TwoSum on the high parts, the small parts in plain
// float64 -- the 13 sums rounded once at the end.  sums15: [0] m, [1..13], [14] sum d2.
// Round 6: the thirteen accumulators take the same steps on different numbers, so they advance together as four vectors of four
// doubles (three lanes idle): element-wise IEEE operations without contraction = the bits of the scalar loop (20 000 random merges
// compared on the host), 0.55 -> 0.13 us per merge of 23 workgroups on the development box -- on the path of every one of a scan's
// ~180 dependent evaluations.  An AVX2 clone where the CPU has it, SSE2 halves otherwise.
typedef double gicp_v4 __attribute__((vector_size(32)));
typedef double gicp_v2 __attribute__((vector_size(16)));
__attribute__((always_inline)) static inline void gicp_merge_body(const double* part, int nblk, double* sums15) {
  auto ld4 = [](const double* p) { gicp_v4 v; __builtin_memcpy(&v, p, 32); return v; };
  auto ld22 = [](const double* a, const double* b) { gicp_v2 x, y; __builtin_memcpy(&x, a, 16); __builtin_memcpy(&y, b, 16); return gicp_v4{x[0], x[1], y[0], y[1]}; };
  double m = 0.0, d2 = 0.0;
  gicp_v4 H[4], L[4];
  for (int v = 0; v < 4; ++v) H[v] = L[v] = gicp_v4{0.0, 0.0, 0.0, 0.0};
  for (int b = 0; b < nblk; ++b, part += kGicpPartialStride) {
    m += part[0];
    d2 += part[16];
    // high parts: values 1..6 at words 1..6, 7..13 at words 8..14; low parts 15..20 at words 17..22, 21..27 at words 24..30
    const gicp_v4 bh[4] = {ld4(part + 1), ld22(part + 5, part + 8), ld4(part + 10), gicp_v4{part[14], 0.0, 0.0, 0.0}};
    const gicp_v4 bl[4] = {ld4(part + 17), ld22(part + 21, part + 24), ld4(part + 26), gicp_v4{part[30], 0.0, 0.0, 0.0}};
    for (int v = 0; v < 4; ++v) {
      const gicp_v4 s = H[v] + bh[v];
      const gicp_v4 bb = s - H[v];
      const gicp_v4 e = (H[v] - (s - bb)) + (bh[v] - bb);
      H[v] = s;
      L[v] = (L[v] + bl[v]) + e;
    }
  }
  sums15[0] = m;
  sums15[14] = d2;
  for (int k = 0; k < 13; ++k) sums15[1 + k] = H[k / 4][k % 4] + L[k / 4][k % 4];
}
#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
static void gicp_merge_blocks(const double* part, int nblk, double* sums15) { gicp_merge_body(part, nblk, sums15); }
static_assert(kGicpPartialStride >= 31, "an answer block holds four lines of seven values + tag");

// ---- the resident evaluation server of a BFGS run (gicp_server_kernel) ---------------------------------------------
// A command is the 12 floats of T, then the sequence number, in one 64-byte line of fine-grained device memory; the device acts
// once the line carries the number it waits for.  (Round 4 measured two variations, neither faster: a line per workgroup --
// written 23 times by the host -- and four reads in flight per workgroup instead of one; per-workgroup stamps show all workgroups
// seeing a command within 0.2 us of each other as it is, a read of the line taking 0.18 us.)
static void gicp_server_command(icpgpu_ctx* c, unsigned int seq, const Xform& T) {
#if defined(__x86_64__)
  // T first, the number last, a store fence in between and behind: posted writes reach the device in that order
  volatile unsigned int* line = c->gicp_cmd;
  for (int k = 0; k < 12; ++k) {
    unsigned int w;
    std::memcpy(&w, &T.m[k], sizeof w);
    line[k] = w;
  }
  _mm_sfence();
  line[12] = seq;
  _mm_sfence();
#else
  (void)c; (void)seq; (void)T;
#endif
}

// Start the server for the evaluations numbered sums_seq + 1, + 2, ... (queued behind whatever the stream still holds).
static int gicp_server_start(icpgpu_ctx* c, int n_s, const unsigned long long* keys, float thr, const Xform& base,
                             const double* maha) {
  c->gicp_server_on = false;
  if (!c->gicp_cmd || !c->gicp_server_allowed) return ICPGPU_OK;
  unsigned int next = (unsigned int)(c->sums_seq + 1);
  if (next == kGicpServerExit || next == 0u) {  // keep the two reserved numbers out of the run's first command
    c->sums_seq += 2;
    next = (unsigned int)(c->sums_seq + 1);
  }
  Xform none{};
  gicp_server_command(c, next - 1u, none);  // a number the server does not wait for: the line may still hold an old exit
  HIP_TRY(c, launch_gicp_server(gicp_direct_blocks(n_s, c->gicp_blocks_most), c->src.data(), n_s, c->tgt.data(), keys, thr, base, maha, c->h_gicp_dev, c->h_gicp_flags_dev,
                                c->gicp_cmd, next, (unsigned int)(wire_seq(c, c->sums_seq + 1) >> 32), c->stream));
  c->gicp_server_on = true;
  return ICPGPU_OK;
}

static void gicp_server_stop(icpgpu_ctx* c) {
  if (!c->gicp_server_on) return;
  Xform none{};
  gicp_server_command(c, kGicpServerExit, none);
  c->gicp_server_on = false;
  // wait for the poller's acknowledgement (a few microseconds): the command line is about to be reused
  for (unsigned spins = 1; c->h_gicp_flags[0] != ~0ull; ++spins) {
    if ((spins & 0x3FFu) == 0 && hipStreamQuery(c->stream) != hipErrorNotReady) break;  // gone already (or an error: the caller's next call reports it)
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}

// 0: the flags arrived; 1: the stream went idle without them (the server gave up waiting); < 0: error
// What a block publishes (icp_kernels.h: four answer lines of seven values + a tag): m, the 13 sums' high parts, sum d2, their low parts.
// (round 6: a line's seven folds are taken four words at a time -- vectors of four 64-bit words, an AVX2 clone where the CPU has it --
//  and every line is looked at before the verdict: 0.37 -> 0.23 us for an evaluation's 92 lines on the development box, the same
//  decisions on 20 000 random mailboxes, torn and stale lines among them; this runs once per dependent evaluation, right after its
//  last answer has arrived)
typedef unsigned long long gicp_u4 __attribute__((vector_size(32)));
__attribute__((always_inline)) static inline bool gicp_tags_ready_body(const double* mailbox, int n_blocks, unsigned long long seq) {
  asm volatile("" ::: "memory");  // the device writes these lines: every call reads them anew
  const unsigned long long* w0 = reinterpret_cast<const unsigned long long*>(mailbox);
  // the numbers first (one compare per line, four lines per workgroup: this loop runs thousands of times per scan), the checksums
  // only once every line carries the number.  The pass over the numbers has NO early exit: the device's writes invalidate the
  // host's cached copies of these lines, and a loop that leaves at the first stale line fetches them one miss after the other
  // -- without the branch the loads are independent and the misses overlap.
  unsigned long long stale = 0;
  const unsigned long long* w = w0;
  for (int b = 0; b < n_blocks; ++b, w += kGicpPartialStride)
    for (int L = 0; L < kGicpLines; ++L) stale |= (w[8 * L + 7] >> 24) ^ seq;
  if (stale) return false;
  w = w0;
  unsigned long long bad = 0;  // (a torn line: looked at again on the next poll)
  const gicp_u4 mask = {0xFFFFFFull, 0xFFFFFFull, 0xFFFFFFull, 0xFFFFFFull}, values3 = {~0ull, ~0ull, ~0ull, 0ull};
  for (int b = 0; b < n_blocks; ++b, w += kGicpPartialStride)
    for (int L = 0; L < kGicpLines; ++L) {
      gicp_u4 lo4, hi4;
      __builtin_memcpy(&lo4, w + 8 * L, 32);
      __builtin_memcpy(&hi4, w + 8 * L + 4, 32);
      const unsigned long long tag = hi4[3];
      hi4 &= values3;  // (the fold of zero is zero)
      const gicp_u4 f = ((lo4 ^ (lo4 >> 24) ^ (lo4 >> 48)) ^ (hi4 ^ (hi4 >> 24) ^ (hi4 >> 48))) & mask;  // gicp_line_fold, word by word
      bad |= tag ^ ((seq << 24) | (f[0] ^ f[1] ^ f[2] ^ f[3]));
    }
  return bad == 0;
}
#if defined(__x86_64__) && defined(__clang__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
static bool gicp_tags_ready(const double* mailbox, int n_blocks, unsigned long long seq) { return gicp_tags_ready_body(mailbox, n_blocks, seq); }
// 0 = all entries of evaluation `seq` are there, 1 = the stream went idle without them (the server gave up), < 0 = error
static int wait_gicp_tags(icpgpu_ctx* c, int n_blocks, unsigned long long seq, bool server) {
  std::chrono::steady_clock::time_point t0;
  static const bool timing = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_TIMING"); return e && std::atoi(e) != 0; }();
  bool any_seen = false;
  std::chrono::steady_clock::time_point t_any;
  for (unsigned spins = 1;; ++spins) {
    if (timing && !any_seen) {  // (development flavour: when does the FIRST workgroup's result show up, when the last?)
      const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(c->h_gicp);
      for (int b = 0; b < n_blocks && !any_seen; ++b) any_seen = (w[(size_t)b * kGicpPartialStride + 8 * (kGicpLines - 1) + 7] >> 24) == seq;
      if (any_seen) t_any = std::chrono::steady_clock::now();
    }
    if (gicp_tags_ready(c->h_gicp, n_blocks, seq)) {
      if (timing && any_seen) c->gt_trickle += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_any).count();
      break;
    }
    if ((spins & 0x3FFu) == 0) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {
        if (gicp_tags_ready(c->h_gicp, n_blocks, seq)) break;
        if (server) return 1;
        return fail(c, ICPGPU_ERR_HIP, "GICP evaluation finished without publishing its result");
      }
      if (q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for a GICP evaluation: %s", hipGetErrorString(q));
      const auto now = std::chrono::steady_clock::now();
      if (spins == 0x400u) t0 = now;
      else if (std::chrono::duration<double, std::milli>(now - t0).count() > wait_timeout_ms())
        return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for a GICP evaluation (hung kernel?)", wait_timeout_ms());
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// The device solver's result: kSolveOut granules carrying number `seq`.  0 = all there (values in out[]), 1 = the stream went
// idle without them (the kernel gave up: the caller falls back to the host's solver), < 0 = error.
static int wait_solve_result(icpgpu_ctx* c, unsigned long long seq, double* out) {
  const int n = gicp_solve_out_granules();
  auto all_there = [&]() {
    bool all = true;
    for (int k = 0; k < n; ++k) all = gicp_granule_read(c->h_solve + 2 * k, seq, &out[k]) && all;
    return all;
  };
  std::chrono::steady_clock::time_point t0;
  for (unsigned spins = 1;; ++spins) {
    if (all_there()) break;
    if ((spins & 0x3FFu) == 0) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {
        if (all_there()) break;
        return 1;
      }
      if (q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for the GICP device solver: %s", hipGetErrorString(q));
      const auto now = std::chrono::steady_clock::now();
      if (spins == 0x400u) t0 = now;
      else if (std::chrono::duration<double, std::milli>(now - t0).count() > wait_timeout_ms())
        return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for the GICP device solver (hung kernel?)", wait_timeout_ms());
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// The quadratic form of an outer iteration (gicp_quadratic_kernel): 2 x kGicpQuadSums result pairs numbered `seq`.  all_there()
// is also what a resumable run polls.
// launch one pass (numbered seq) on the context's stream
static int quad_pass_launch(icpgpu_ctx* c, int n_s, const unsigned long long* keys, float thr_excl, const Rot3d& R, unsigned long long seq) {
  if (c->quad_pending) HIP_TRY(c, hipMemsetAsync(c->quad_done, 0, sizeof(unsigned int), c->stream));  // the last pass never reported: its counter may be mid-count
  c->quad_pending = true;
  HIP_TRY(c, launch_gicp_quadratic(c->src.data(), n_s, c->tgt.data(), keys, thr_excl, R, static_cast<const double*>(c->cov_src.ptr),
                                   static_cast<const double*>(c->cov_tgt.ptr), c->quad_partials, c->quad_done, c->h_quad_dev,
                                   wire_seq(c, seq), c->stream));
  return ICPGPU_OK;
}
static bool quad_sums_read(const icpgpu_ctx* c, unsigned long long seq, double* sums) {
  bool all = true;
  for (int k = 0; k < 2 * kGicpQuadSums; ++k) all = gicp_granule_read(c->h_quad + 2 * k, seq, &sums[k]) && all;
  return all;
}
static int wait_quad_sums(icpgpu_ctx* c, unsigned long long seq, double* sums) {
  std::chrono::steady_clock::time_point t0;
  for (unsigned spins = 1;; ++spins) {
    // (the last pair the kernel stores first: nothing else needs looking at until it is there)
    double probe;
    if (gicp_granule_read(c->h_quad + 2 * (2 * kGicpQuadSums - 1), seq, &probe) && quad_sums_read(c, seq, sums)) break;
    if ((spins & 0x3FFu) == 0) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {
        if (quad_sums_read(c, seq, sums)) break;
        return fail(c, ICPGPU_ERR_HIP, "the GICP quadratic pass finished without publishing its sums");
      }
      if (q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for the GICP quadratic pass: %s", hipGetErrorString(q));
      const auto now = std::chrono::steady_clock::now();
      if (spins == 0x400u) t0 = now;
      else if (std::chrono::duration<double, std::milli>(now - t0).count() > wait_timeout_ms())
        return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for the GICP quadratic pass (hung kernel?)", wait_timeout_ms());
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// workgroups of a device-solver run: one per 1024 correspondences while every lane's share stays one quad in registers; larger
// clouds stream their shares with at most kSolveStreamBlocks workgroups (every workgroup gathers every other's 28 granules per
// evaluation: the gather grows with the count)
static int gicp_solve_blocks(int n_s, int most) {
  constexpr int kSolveStreamBlocks = 64;
  static const int cap = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_SOLVE_BLOCKS"); const int v = e ? std::atoi(e) : kSolveStreamBlocks; return v < 1 ? 1 : (v > kGicpDirectBlocks ? kGicpDirectBlocks : v); }();
  int blocks = (n_s + 1023) / 1024;
  if (blocks > cap) blocks = cap;
  if (blocks > most) blocks = most;
  return blocks < 1 ? 1 : blocks;
}

// PCL's convergence measure of an outer iteration: the largest entry of |previous - transformation|, rotation entries weighed by
// 1 / rotation_epsilon_, the others by 1 / transformation_epsilon_
static double gicp_outer_delta(const float previous[16], const float transformation[16], double rot_eps, double trans_eps) {
  double delta = 0.0;
  for (int k = 0; k < 4; ++k)
    for (int l = 0; l < 4; ++l) {
      const double ratio = (k < 3 && l < 3) ? 1.0 / rot_eps : 1.0 / trans_eps;
      delta = std::max(delta, ratio * std::fabs((double)previous[l * 4 + k] - (double)transformation[l * 4 + k]));
    }
  return delta;
}
// PCL's own composition of the result: R = previous.R * guess.R, t = previous.t + guess.t
static void gicp_compose_final(const float previous[16], const float guess[16], float fin[16]) {
  mat4f_identity(fin);
  for (int r = 0; r < 3; ++r) {
    for (int cc = 0; cc < 3; ++cc) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += previous[k * 4 + r] * guess[cc * 4 + k];
      fin[cc * 4 + r] = s;
    }
    fin[12 + r] = previous[12 + r] + guess[12 + r];
  }
}

int align_gicp(icpgpu_ctx* c, const float* guess_in, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  struct ServerGuard {  // whatever way this function is left, no server stays behind
    icpgpu_ctx* c;
    ~ServerGuard() { gicp_server_stop(c); }
  } server_guard{c};
  struct SpecGuard {  // ... and no covariance grid whose statistics nobody has looked at
    icpgpu_ctx* c;
    ~SpecGuard() {
      if (c->spec_grid.pending) (void)covariance_grid_check(c);
    }
  } spec_guard{c};
  const auto t_start = std::chrono::steady_clock::now();
  // development flavour, ICPGPU_GICP_TIMING=1: host wall per stage (printed by icpgpu_destroy)
  static const bool stage_timing = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_TIMING"); return e && std::atoi(e) != 0; }();
  auto t_mark = t_start;
  auto mark = [&](int stage) {
    if (!stage_timing) return;
    const auto now = std::chrono::steady_clock::now();
    c->gt_stage[stage] += std::chrono::duration<double, std::micro>(now - t_mark).count();
    t_mark = now;
  };
  init_result(res);
  {
    const int grc = ensure_gicp_resources(c);
    if (grc) return grc;
  }
  c->prof.aligns += 1;
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  const icpgpu_params& P = c->params;
  const bool quadratic = gicp_inner_quadratic(c);  // the inner minimisation on the quadratic form (icp_gicp_quadratic.h)
  if (quadratic) {
    const int qrc = ensure_gicp_quadratic_resources(c);
    if (qrc) return qrc;
  }
  c->prev.valid = c->tile_seed.valid = false;  // every alignment starts cold
  float guess[16];
  if (guess_in) std::memcpy(guess, guess_in, sizeof(guess));
  else mat4f_identity(guess);

  auto finish_early = [&]() {  // empty target / clouds smaller than k_correspondences_: PCL leaves converged_ = false, T = I
    c->final_T = mat4_identity();
    c->have_final = true;
    int rc = write_output_cloud(c, to_xform(c->final_T), out_xyzw);
    res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    return rc;
  };
  if (n_t == 0 || n_s < kGicpK || n_t < kGicpK) return finish_early();

  int rc;
  if ((rc = ensure_covariances(c, c->tgt, c->tgt_version, c->cov_grid_tgt, c->cov_tgt, c->cov_tgt_version))) return rc;
  if ((rc = ensure_covariances(c, c->src, c->src_version, c->cov_grid_src, c->cov_src, c->cov_src_version, /*allow_unchecked=*/true))) return rc;
  mark(0);
  // GICP keeps d2 < r^2 (strict): the largest float below r^2
  const double r2 = P.max_correspondence_distance * P.max_correspondence_distance;
  float thr = threshold_from(r2);
  if ((double)thr >= r2) thr = std::nextafterf(thr, -INFINITY);
  const float thr_excl = std::nextafterf(thr, INFINITY);  // d2 < thr_excl  <=>  d2 <= thr
  if ((rc = ensure_grid(c, thr))) return rc;
  if ((rc = ensure(c, c->keys, (size_t)n_s * sizeof(unsigned long long)))) return rc;
  if ((rc = ensure(c, c->maha, (size_t)n_s * 6 * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double)))) return rc;
  auto* keys = static_cast<unsigned long long*>(c->keys.ptr);
  auto* maha = static_cast<double*>(c->maha.ptr);
  const Xform base = xform_from_f16(guess);
  mark(1);

  float transformation[16], previous[16];
  mat4f_identity(transformation);
  mat4f_identity(previous);
  int nr = 0, state = ICPGPU_NOT_CONVERGED;
  bool converged = false;
  unsigned n_corr = 0;
  double mse = 0.0, dev_ms = 0.0;
  const double rot_eps = 2e-3;  // PCL rotation_epsilon_ (never set by the reference)

  while (!converged) {
    float TG[16];
    mat4f_mul(transformation, guess, TG);
    const Xform Tq = xform_from_f16(TG);
    Rot3d R;
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += (double)transformation[k * 4 + r] * (double)guess[cc * 4 + k];
        R.m[3 * r + cc] = s;
      }
    // correspondences: exact NN keys (only those with d2 < r^2 are used, so the grid's cutoff search is complete)
    // (timed like the point-to-point sweeps: one outer iteration in `timing_every` -- two event records are barrier packets in
    //  front of and behind the search, a few microseconds of every outer iteration when each is timed)
    const bool timed = c->timing_every <= 1 || (c->sweep_counter++ % (unsigned)c->timing_every) == 0;
    if (timed) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    if (grid_ready(c)) {
      unsigned int* prev = nullptr;  // each outer iteration's neighbours bound the next one's search
      bool use_prev = false;
      int prc = prev_neighbours(c, c->grid, c->src.data(), n_s, grid_flags(c->grid, false), prev, use_prev);
      if (prc) return prc;
      HIP_TRY(c, launch_nn_grid_search(c->src.data(), n_s, grid_flags(c->grid, false), Tq, static_cast<const float4*>(c->grid.sorted.ptr),
                                       static_cast<const int*>(c->grid.cell_start.ptr), c->grid.g, thr, keys, nullptr, nullptr,
                                       nullptr, c->stream, prev, use_prev));
    } else {
      if ((rc = nn_keys_brute(c, c->tgt.data(), n_t, Tq, keys))) return rc;
    }
    if (!quadratic)
      HIP_TRY(c, launch_gicp_mahalanobis(n_s, keys, thr_excl, R, static_cast<const double*>(c->cov_src.ptr),
                                         static_cast<const double*>(c->cov_tgt.ptr), maha, c->stream));
    if (timed) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    mark(2);
    // the source's covariance grid was built ahead of its statistics (ensure_covariances): they have arrived by now -- the covariance
    // pass, this search and the Mahalanobis kernel were queued behind them.  A grid that failed (a point outside the box it was
    // given) is rebuilt the waiting way and the alignment starts over; nothing of it has reached the host yet.
    if (c->spec_grid.pending) {
      const int chk = covariance_grid_check(c);
      if (chk < 0) return chk;
      if (chk == 1) {
        c->prof.aligns -= 1;
        return align_gicp(c, guess_in, out_xyzw, want_fitness, res);
      }
    }

    // rigid_transformation_estimation_: BFGS over x = (t, roll, pitch, yaw), every evaluation one reduction on the device
    double m_count = 0.0;
    auto eval = [&](const Vec6& x, GicpEval& out) -> bool {
      float T[16];
      std::memcpy(T, guess, sizeof(T));
      const gicp::Trig6 tr = gicp::trig6(x);  // the state's six sine / cosine pairs (correctly rounded, double-double: ~0.2 us on the
      gicp::apply_state(T, x, tr);            //  host) once per evaluation: the transform here, the gradient below (as the device solver does)
      // ~300 evaluations per align, each a dependent launch: ONE kernel of a few workgroups whose partial sums land in the
      // polled host mailbox; the host adds them in workgroup order (deterministic)
      const auto t_eval0 = std::chrono::steady_clock::now();
      unsigned long long seq = ++c->sums_seq;
      if ((unsigned int)seq == kGicpServerExit) seq = (c->sums_seq += 2);  // (never a command number; the server skips it too)
      const int nblk = gicp_direct_blocks(n_s, c->gicp_blocks_most);
      bool have = false;
      static const bool timing = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_TIMING"); return e && std::atoi(e) != 0; }();
      std::chrono::steady_clock::time_point tq0, tq1, tq2;
      if (timing) {
        tq0 = std::chrono::steady_clock::now();
        if (c->gt_n && c->gicp_server_on) c->gt_between += std::chrono::duration<double, std::micro>(tq0 - c->gt_last).count();
        if (c->gt_pending > 0) {  // the last evaluation's device stamps (icp_gicp.hip: gicp_server_kernel), 100 MHz ticks
          double first_seen = 1e300, acc = 0, red = 0, poll = 0;
          for (int b = 0; b < c->gt_pending; ++b) {
            const double* o = c->h_gicp + (size_t)b * kGicpPartialStride;
            const double seen = o[2 * 29], t_acc = o[2 * 30], done = o[2 * 31], loop = o[2 * 29 + 1];
            first_seen = std::min(first_seen, seen);
            acc += t_acc - seen;
            red += done - t_acc;
            poll += seen - loop;
          }
          if (c->gt_dev_n >= 1000 && c->gt_dev_n < 1003) {  // a few evaluations in full
            fprintf(stderr, "[icpgpu] evaluation %llu, per workgroup (us after the first one saw the command): xcd | started polling | saw it | accumulated | stored | reads\n", c->gt_dev_n);
            for (int b = 0; b < c->gt_pending; ++b) {
              const double* o = c->h_gicp + (size_t)b * kGicpPartialStride;
              fprintf(stderr, "[icpgpu]   %2d: %d | %7.2f | %5.2f | %5.2f | %5.2f | %3.0f\n", b, (int)o[2 * 31 + 1], (o[2 * 29 + 1] - first_seen) * 0.01,
                      (o[2 * 29] - first_seen) * 0.01, (o[2 * 30] - first_seen) * 0.01, (o[2 * 31] - first_seen) * 0.01, o[2 * 30 + 1]);
            }
          }
          c->gt_dev_wait += poll / c->gt_pending * 0.01;
          c->gt_dev_work += acc / c->gt_pending * 0.01;
          c->gt_dev_reduce += red / c->gt_pending * 0.01;
          c->gt_dev_n += 1;
          c->gt_pending = 0;
        }
      }
      if (c->gicp_server_on) {  // the resident server evaluates; no launch
        gicp_server_command(c, (unsigned int)seq, xform_from_f16(T));
        if (timing) tq1 = std::chrono::steady_clock::now();
        const int w = wait_gicp_tags(c, nblk, seq, /*server=*/true);
        if (timing) tq2 = std::chrono::steady_clock::now();
        if (w < 0) return false;
        have = w == 0;
        if (!have) c->gicp_server_on = false;  // it gave up (50 ms without a command): single launches from here on
        if (!have && std::getenv("ICPGPU_DEBUG")) fprintf(stderr, "[icpgpu] gicp server gave up at evaluation %llu\n", seq);
      }
      if (!have) {
        if (launch_gicp_cost_direct(nblk, c->src.data(), n_s, c->tgt.data(), keys, thr_excl, xform_from_f16(T), base, maha,
                                    c->h_gicp_dev, c->h_gicp_flags_dev, wire_seq(c, seq), c->stream) != hipSuccess)
          return false;
        if (wait_gicp_tags(c, nblk, seq, /*server=*/false) != 0) return false;
      }
      gicp_merge_blocks(c->h_gicp, nblk, c->h_sums);  // workgroup by workgroup, in double-double: the 13 sums are rounded once, here
      if (timing && have) {
        const auto tq3 = std::chrono::steady_clock::now();
        c->gt_cmd += std::chrono::duration<double, std::micro>(tq1 - tq0).count();
        c->gt_wait += std::chrono::duration<double, std::micro>(tq2 - tq1).count();
        c->gt_merge += std::chrono::duration<double, std::micro>(tq3 - tq2).count();
        c->gt_pending = nblk;  // the workgroups' stamps of this evaluation are read when the next one starts (they trail the tags)
        c->gt_n += 1;
        c->gt_last = tq3;
      }
      c->prof.gicp_cost_launches += 1;
      c->prof.gicp_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_eval0).count();
      const double* s = c->h_sums;
      c->prof.gicp_eval_corr += (uint64_t)s[0];
      m_count = s[0];
      mse = s[0] > 0 ? s[14] / s[0] : 0.0;
      gicp::eval_from_sums(tr, s, out);
      return true;
    };
    // rigid_transformation_estimation_: the whole BFGS run on the device (gicp_solve_kernel), one result for the host to poll
    Vec6 x = gicp_state_from_matrix(transformation);
    bool solved = false, leave_outer_loop = false;
    if (quadratic) {
      // ... or on the host over the quadratic form: ONE pass over the correspondences (Mahalanobis matrices on the way), 150
      // numbers to wait for, then BFGS without a device round trip
      const auto t_q0 = std::chrono::steady_clock::now();
      const unsigned long long seq = ++c->quad_seq;
      if ((rc = quad_pass_launch(c, n_s, keys, thr_excl, R, seq))) return rc;
      double sums[2 * kGicpQuadSums];
      if ((rc = wait_quad_sums(c, seq, sums))) return rc;
      c->quad_pending = false;
      mark(3);
      if (stage_timing) {  // the last workgroup's stamps (gicp_quadratic_kernel), 100 MHz ticks
        double st[5];
        bool all = true;
        for (int u = 0; u < 5; ++u) all = gicp_granule_read(c->h_quad + 2 * (2 * kGicpQuadSums + u), seq, &st[u]) && all;
        if (all) {
          long long t[5];
          for (int u = 0; u < 5; ++u) std::memcpy(&t[u], &st[u], sizeof(long long));
          static double acc[4] = {0, 0, 0, 0};
          static unsigned long long n = 0;
          for (int u = 0; u < 4; ++u) acc[u] += (double)(t[u + 1] - t[u]) * 0.01;
          if ((++n % 100) == 0)
            fprintf(stderr, "[icpgpu] quadratic pass, last workgroup, mean of %llu (us): loads + Mahalanobis %.2f | six groups %.2f | fence + counter %.2f | partials of all workgroups %.2f\n",
                    n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n);
        }
      }
      const double m = sums[2 * 73], d2 = sums[2 * 74];
      m_count = m;
      mse = m > 0 ? d2 / m : 0.0;
      n_corr = (unsigned)m;
      std::memcpy(previous, transformation, sizeof(previous));
      if (n_corr < 4) {  // NotEnoughPointsException -> the loop breaks with converged_ = false
        state = ICPGPU_CONV_NO_CORRESPONDENCES;
        break;
      }
      int evals = 0;
      const GicpSolve sr = gicp_minimize_quadratic(sums, guess, x, 20, 1e-2, &evals);
      mark(4);
      c->prof.gicp_quadratic_solves += 1;
      res->gicp_solver = ICPGPU_GICP_SOLVER_QUADRATIC;
      c->prof.gicp_cost_launches += (uint64_t)evals;
      c->prof.gicp_eval_corr += (uint64_t)(m * evals);
      c->prof.gicp_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_q0).count();
      if (sr != GicpSolve::Ok) {  // SolverDidntConvergeException
        state = ICPGPU_NOT_CONVERGED;
        break;
      }
      solved = true;
    }
    // which solver: forced by ICPGPU_GICP_DEVICE, or (default) the one this context has measured to be faster on this box --
    // until it knows, inner minimisations alternate and are timed (same bits either way, so nothing but time depends on it)
    bool try_device = !quadratic && c->gicp_device_ok && c->gicp_server_allowed;
    bool timing_this_run = false;
    if (try_device && gicp_device_solver_mode() == 2) {
      const int nblk = gicp_solve_blocks(n_s, c->gicp_blocks_most);
      const bool fits_one_xcd = c->gicp_local_ok && nblk <= gicp_solve_local_blocks() && (long long)nblk * 1024 >= n_s && 8 * nblk <= c->gicp_blocks_most;
      if (!fits_one_xcd) try_device = false;  // (streamed shares, 13 us gathers: the device solver is 30 % slower there)
      else if (c->gicp_choice == 0) {
        timing_this_run = true;
        try_device = c->gicp_cal_runs[1] <= c->gicp_cal_runs[0];  // the one that has run less
      } else {
        try_device = c->gicp_choice == 2;
      }
    }
    const auto t_inner0 = std::chrono::steady_clock::now();
    const uint64_t evals_before = c->prof.gicp_cost_launches;
    for (int attempt = 0; attempt < 2 && !solved && try_device && c->gicp_device_ok && c->gicp_server_allowed; ++attempt) {
      const auto t_solve0 = std::chrono::steady_clock::now();
      const int nblk = gicp_solve_blocks(n_s, c->gicp_blocks_most);
      // the one-XCD variant when the run fits one XCD's 32 CUs with its correspondences in registers (the reference's
      // voxel-filtered scans do: ~22k points); it needs 8 x nblk workgroups launched, so the context must own the chip's share
      const bool local = c->gicp_local_ok && nblk <= gicp_solve_local_blocks() && (long long)nblk * 1024 >= n_s &&
                         8 * nblk <= c->gicp_blocks_most;
      const unsigned long long seq0 = (c->gicp_solve_seq += 8192);  // evaluation e of the run carries seq0 + e (e < 8192: <= 20 steps of <= 200 trials)
      HIP_TRY(c, launch_gicp_solve(nblk, c->src.data(), n_s, c->tgt.data(), keys, thr_excl, base, guess, maha, x.v, c->gicp_slots,
                                   c->h_solve_dev, wire_seq(c, seq0), 20, 1e-2, c->stream, local ? c->gicp_slots_local : nullptr,
                                   local ? c->gicp_owner : nullptr, c->gicp_xcc));
      double out[24];
      const int w = wait_solve_result(c, seq0, out);
      if (w < 0) return w;
      const int status = w == 0 ? (int)out[0] : (int)gicp::kDeviceError;
      if (status == gicp::kDeviceError) {  // a gather timed out (or the kernel never answered)
        if (local) c->gicp_local_ok = false;  // placement was not what the one-XCD variant needs: the any-placement variant from here on
        else c->gicp_device_ok = false;       // ... and if that one fails too: the host's solver
        if (std::getenv("ICPGPU_DEBUG"))
          fprintf(stderr, "[icpgpu] gicp device solver%s gave up (%s, %d workgroups, record %.0f); falling back\n", local ? " (one XCD)" : "",
                  w == 0 ? "a gather timed out" : "no answer", nblk, w == 0 ? out[11] : -1.0);
        if (std::getenv("ICPGPU_DEBUG") && w == 0)
          fprintf(stderr, "[icpgpu]   seq0 %llu: hi granule number %.0f checksum %.0f, lo granule number %.0f, checksum of the hi bits %.0f\n", seq0, out[12], out[13], out[14], out[15]);
      } else {
        solved = true;
        c->prof.gicp_device_solves += 1;
        res->gicp_solver = ICPGPU_GICP_SOLVER_DEVICE;
        const double m = out[7], evals = out[10];
        m_count = m;
        mse = m > 0 ? out[8] / m : 0.0;
        n_corr = (unsigned)m;
        c->prof.gicp_cost_launches += (uint64_t)evals;
        c->prof.gicp_eval_corr += (uint64_t)(m * evals);
        c->prof.gicp_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_solve0).count();
        std::memcpy(previous, transformation, sizeof(previous));
        if (status == gicp::kNotEnoughPoints) {  // NotEnoughPointsException -> the loop breaks with converged_ = false
          state = ICPGPU_CONV_NO_CORRESPONDENCES;
          leave_outer_loop = true;
          break;
        }
        if (status != gicp::kOk) {  // SolverDidntConvergeException
          state = ICPGPU_NOT_CONVERGED;
          leave_outer_loop = true;
          break;
        }
        for (int k = 0; k < 6; ++k) x[k] = out[1 + k];
      }
    }
    if (leave_outer_loop) break;  // (the two exits above leave the attempt loop only: until round 4's last day the outer loop went on
                                  //  and reported a converged alignment with the unchanged transform -- found when the device
                                  //  solver became part of the default path; tests/test_gpu_gicp.py now runs the degenerate cases through it)
    if (!solved) {
      // the ~35 dependent evaluations of this outer iteration go to a resident kernel (queued behind the two kernels above)
      if ((rc = gicp_server_start(c, n_s, keys, thr_excl, base, maha))) return rc;
      // number of correspondences (one evaluation at the current state; it is also the BFGS start, cached by the solver)
      x = gicp_state_from_matrix(transformation);
      GicpEval probe;
      if (!eval(x, probe)) return fail(c, ICPGPU_ERR_HIP, "GICP cost evaluation failed: %s", hipGetErrorString(hipGetLastError()));
      mark(3);
      n_corr = (unsigned)m_count;
      std::memcpy(previous, transformation, sizeof(previous));
      if (n_corr < 4) {  // NotEnoughPointsException -> the loop breaks with converged_ = false
        state = ICPGPU_CONV_NO_CORRESPONDENCES;
        break;
      }
      const GicpSolve sr = gicp_minimize(eval, x, 20, 1e-2, &probe);
      c->prof.gicp_host_solves += 1;
      res->gicp_solver = ICPGPU_GICP_SOLVER_HOST;
      mark(4);
      gicp_server_stop(c);
      mark(5);
      if (sr == GicpSolve::DeviceError) return fail(c, ICPGPU_ERR_HIP, "GICP cost evaluation failed: %s", hipGetErrorString(hipGetLastError()));
      if (sr != GicpSolve::Ok) {  // SolverDidntConvergeException
        state = ICPGPU_NOT_CONVERGED;
        break;
      }
    }
    if (timing_this_run) {
      const int which = solved ? 1 : 0;
      if (c->gicp_cal_runs[which]++ > 0) {  // (a solver's first run pays for code upload and first-touch: not counted)
        c->gicp_cal_us[which] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_inner0).count();
        c->gicp_cal_evals[which] += c->prof.gicp_cost_launches - evals_before;
      }
      if (!c->gicp_device_ok || (!c->gicp_local_ok && !solved)) {
        c->gicp_choice = 1;  // the device solver gave up on this context
      } else if (c->gicp_cal_evals[0] >= 150 && c->gicp_cal_evals[1] >= 150) {
        const double host_us = c->gicp_cal_us[0] / (double)c->gicp_cal_evals[0], dev_us = c->gicp_cal_us[1] / (double)c->gicp_cal_evals[1];
        c->gicp_choice = dev_us < 0.97 * host_us ? 2 : 1;  // (the host loop on a tie: it is the simpler machine)
        if (std::getenv("ICPGPU_DEBUG"))
          fprintf(stderr, "[icpgpu] GICP inner solver measured on this context: host loop %.2f us, device solver %.2f us per evaluation -> %s\n", host_us,
                  dev_us, c->gicp_choice == 2 ? "device solver" : "host loop");
      }
    }
    mat4f_identity(transformation);
    gicp_apply_state(transformation, x);
    float ms = 0.f;
    if (timed) HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    dev_ms += ms;
    c->prof.grid_launches += grid_ready(c) ? 1 : 0;
    c->prof.grid_ms += grid_ready(c) ? ms : 0.0;
    c->prof.grid_timed += (timed && grid_ready(c)) ? 1 : 0;
    mark(6);
    const double delta = gicp_outer_delta(previous, transformation, rot_eps, P.transformation_epsilon);
    ++nr;
    c->prof.iterations += 1;
    if (nr >= P.max_iterations || (delta < 1 && !P.force_iterations)) {
      converged = true;
      state = nr >= P.max_iterations ? ICPGPU_CONV_ITERATIONS : ICPGPU_CONV_TRANSFORM;
      std::memcpy(previous, transformation, sizeof(previous));
    }
  }
  float fin[16];
  gicp_compose_final(previous, guess, fin);
  std::memcpy(res->T, fin, sizeof(fin));
  for (int i = 0; i < 16; ++i) c->final_T[i] = (double)fin[i];
  c->have_final = true;
  res->converged = converged ? 1 : 0;
  res->iterations = nr;
  res->convergence_state = state;
  res->n_correspondences = n_corr;
  res->mse_last = mse;
  const Xform Tf = xform_from_f16(fin);
  // The aligned cloud and the fitness the reference asks for on every scan (icp_odometer.cpp:196-201), round 6: the transform kernel
  // writes the cloud into the pinned staging buffer, a marker follows, the fitness sweep is queued behind both -- and the host copies
  // the cloud out while the sweep runs (it used to wait for the sweep, then for a transform, a copy engine and a marker, then copy).
  StageTicket out_ticket;
  bool out_done = false;
  if (want_fitness) {
    if ((rc = resolve_sweep_timings(c))) return rc;
    c->dev_ms_accum = 0.0;
    if ((rc = output_cloud_issue(c, Tf, out_xyzw, out_ticket))) return rc;
    SweepTicket tk;
    if ((rc = sweep_issue(c, Tf, FLT_MAX, true, tk))) return rc;
    if (out_ticket.issued) {
      if ((rc = output_cloud_complete(c, out_ticket, out_xyzw))) return rc;
      out_done = true;
    }
    if ((rc = wait_sums(c, tk.seq))) return rc;
    if ((rc = sweep_complete(c, tk))) return rc;
    if ((rc = resolve_sweep_timings(c))) return rc;
    dev_ms += c->dev_ms_accum;  // 0 when this sweep was not a timed one
    res->fitness = c->h_sums[0] > 0.0 ? c->h_sums[16] / c->h_sums[0] : DBL_MAX;
  }
  mark(7);
  if (!out_done && (rc = write_output_cloud(c, Tf, out_xyzw))) return rc;
  mark(8);
  if ((rc = resolve_cov_timing(c))) return rc;  // (long finished: the evaluations ran behind it)
  if (stage_timing) c->gt_aligns += 1;
  res->t_device_ms = dev_ms;
  res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  return ICPGPU_OK;
}


// ---- a GICP registration as a RESUMABLE run (icpgpu_align_batch: one host thread keeps several in flight) -------------------
// align_gicp above is a blocking host loop: the index builds wait for their read-backs, every inner minimisation is ~30 dependent
// host <-> device round trips (or one wait for the device solver).  A GicpRun is the same registration cut at its waits: the
// covariance grids go through their host round trips as posted markers the host POLLS, every outer iteration is search +
// Mahalanobis + the whole BFGS run in ONE resident kernel (gicp_solve_kernel, as align_gicp's device-solver branch launches it),
// whose result granules the host polls, and the fitness sweep is a SweepTicket like point-to-point's.  Same kernels, same
// arguments, same host arithmetic between them as align_gicp => the same bits (tests/test_gpu_gicp.py compares).
// What a run cannot take -- a guess or an output cloud (the batch passes neither), a context without the device solver, a solver
// that gives up (a gather timed out: co-residency lost) -- goes through align_gicp: gicp_run_begin says so (phase Blocking) and
// gicp_run_step then runs the blocking function; results do not depend on the path, so a restart from scratch is safe.
static unsigned long long run_post_marker(icpgpu_ctx* c, const int* d_any, int* rc) {
  const unsigned long long number = ++c->post_seq;
  *rc = ICPGPU_OK;
  if (launch_post_ints(d_any, 1, c->h_post_dev, wire_seq(c, number), c->stream) != hipSuccess) *rc = fail(c, ICPGPU_ERR_HIP, "GICP run: marker launch");
  return number;
}
// (>=: markers and fetch_ints' posts are numbered in the order they were queued on the context's one stream -- a later one showing
//  means the awaited one has been written too)
static bool run_marker_seen(const icpgpu_ctx* c, unsigned long long number) { return (c->h_post[1] >> 24) >= number; }

static bool gicp_run_device_ok(const icpgpu_ctx* c) {
  return gicp_device_solver_mode() != 0 && c->gicp_device_ok && c->gicp_server_allowed && c->gicp_slots && c->h_solve;
}

static int gicp_run_finish_early(icpgpu_ctx* c, GicpRun& r) {  // empty target / clouds smaller than k_correspondences_
  c->final_T = mat4_identity();
  c->have_final = true;
  r.res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_start).count();
  r.phase = GicpRun::Done;
  return ICPGPU_OK;
}

// queue one outer iteration: correspondences, Mahalanobis matrices, the BFGS run
static int gicp_run_queue_outer(icpgpu_ctx* c, GicpRun& r) {
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  auto* keys = static_cast<unsigned long long*>(c->keys.ptr);
  auto* maha = static_cast<double*>(c->maha.ptr);
  float TG[16];
  mat4f_mul(r.transformation, r.guess, TG);
  const Xform Tq = xform_from_f16(TG);
  Rot3d R;
  for (int rr = 0; rr < 3; ++rr)
    for (int cc = 0; cc < 3; ++cc) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)r.transformation[k * 4 + rr] * (double)r.guess[cc * 4 + k];
      R.m[3 * rr + cc] = s;
    }
  int rc;
  if (grid_ready(c)) {
    unsigned int* prev = nullptr;  // each outer iteration's neighbours bound the next one's search
    bool use_prev = false;
    if ((rc = prev_neighbours(c, c->grid, c->src.data(), n_s, grid_flags(c->grid, false), prev, use_prev))) return rc;
    HIP_TRY(c, launch_nn_grid_search(c->src.data(), n_s, grid_flags(c->grid, false), Tq, static_cast<const float4*>(c->grid.sorted.ptr),
                                     static_cast<const int*>(c->grid.cell_start.ptr), c->grid.g, r.thr, keys, nullptr, nullptr,
                                     nullptr, c->stream, prev, use_prev));
    c->prof.grid_launches += 1;
  } else {
    if ((rc = nn_keys_brute(c, c->tgt.data(), n_t, Tq, keys))) return rc;
  }
  if (r.quadratic) {  // one pass for the quadratic form's sums (Mahalanobis matrices on the way); the run polls them (phase Quad)
    r.seq0 = ++c->quad_seq;
    r.t_issue = std::chrono::steady_clock::now();
    if ((rc = quad_pass_launch(c, n_s, keys, r.thr_excl, R, r.seq0))) return rc;
    r.solve_stream = c->stream;
    r.phase = GicpRun::Quad;
    r.polls = 0;
    return ICPGPU_OK;
  }
  HIP_TRY(c, launch_gicp_mahalanobis(n_s, keys, r.thr_excl, R, static_cast<const double*>(c->cov_src.ptr),
                                     static_cast<const double*>(c->cov_tgt.ptr), maha, c->stream));
  Vec6 x = gicp_state_from_matrix(r.transformation);
  const int nblk = gicp_solve_blocks(n_s, c->gicp_blocks_most);
  if (r.combine && (long long)nblk * 1024 >= n_s) {  // the scheduler launches it together with the other runs that are ready
    GicpSolveItem& it = r.item;
    it.src = c->src.data();
    it.n_s = n_s;
    it.tgt = c->tgt.data();
    it.keys = keys;
    it.thr = r.thr_excl;
    it.base = xform_from_f16(r.guess);
    std::memcpy(it.guess, r.guess, sizeof(it.guess));
    it.maha6 = maha;
    for (int k = 0; k < 6; ++k) it.x0[k] = x.v[k];
    it.slots = c->gicp_slots;
    it.host_out = c->h_solve_dev;
    r.seq0 = (c->gicp_solve_seq += 8192);
    it.seq0 = wire_seq(c, r.seq0);
    it.blocks = nblk;
    r.local = false;
    r.phase = GicpRun::WantSolve;
    return ICPGPU_OK;
  }
  // the one-XCD variant wherever the run fits one XCD with its correspondences in registers (align_gicp also asks for 8 x nblk <=
  // the context's share of the chip: a lone blocking alignment sizes its evaluation server by that share; the runs of a batch
  // sit on their contexts' own XCDs, c->gicp_xcc, and leave after every outer iteration)
  r.local = c->gicp_local_ok && nblk <= gicp_solve_local_blocks() && (long long)nblk * 1024 >= n_s;
  r.seq0 = (c->gicp_solve_seq += 8192);
  r.t_issue = std::chrono::steady_clock::now();
  HIP_TRY(c, launch_gicp_solve(nblk, c->src.data(), n_s, c->tgt.data(), keys, r.thr_excl, xform_from_f16(r.guess), r.guess, maha, x.v, c->gicp_slots,
                               c->h_solve_dev, wire_seq(c, r.seq0), 20, 1e-2, c->stream, r.local ? c->gicp_slots_local : nullptr,
                               r.local ? c->gicp_owner : nullptr, c->gicp_xcc));
  r.solve_stream = c->stream;
  r.phase = GicpRun::Solve;
  r.polls = 0;
  return ICPGPU_OK;
}

void gicp_run_solver_launched(icpgpu_ctx* c, GicpRun& r, hipStream_t solve_stream) {
  (void)c;
  r.solve_stream = solve_stream;
  r.t_issue = std::chrono::steady_clock::now();
  r.phase = GicpRun::Solve;
  r.polls = 0;
}

// after the covariances: the search grid, the scratch, the first outer iteration
static int gicp_run_start_outer(icpgpu_ctx* c, GicpRun& r) {
  const int n_s = (int)c->src.n;
  int rc;
  if ((rc = ensure_grid(c, r.thr))) return rc;  // (GICP: adopts the grid the target's covariances were computed over -- no build)
  if ((rc = ensure(c, c->keys, (size_t)n_s * sizeof(unsigned long long)))) return rc;
  if ((rc = ensure(c, c->maha, (size_t)n_s * 6 * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double)))) return rc;
  return gicp_run_queue_outer(c, r);
}

// the next covariance grid that needs building (target first, then source), or on to the outer iterations
static int gicp_run_next_cov(icpgpu_ctx* c, GicpRun& r) {
  int rc;
  while (r.cov_stage < 2) {
    const bool tgt = r.cov_stage == 0;
    const Cloud& cloud = tgt ? c->tgt : c->src;
    GridIndex& G = tgt ? c->cov_grid_tgt : c->cov_grid_src;
    DeviceBuf& cov = tgt ? c->cov_tgt : c->cov_src;
    uint64_t& cov_version = tgt ? c->cov_tgt_version : c->cov_src_version;
    const uint64_t version = tgt ? c->tgt_version : c->src_version;
    bool needed = false;
    r.gb = GridBuild{};
    if ((rc = cov_grid_begin(c, cloud, version, G, cov, cov_version, r.gb, needed))) return rc;
    if (needed && r.gb.state != GridBuild::Done) {  // a read-back is queued: poll the marker behind it
      r.marker = run_post_marker(c, static_cast<const int*>(G.ints.ptr), &rc);
      if (rc) return rc;
      r.phase = GicpRun::CovGrid;
      r.polls = 0;
      r.t_issue = std::chrono::steady_clock::now();
      return ICPGPU_OK;
    }
    if (needed && (rc = cov_launch(c, cloud, version, G, cov, cov_version, /*timed=*/false))) return rc;
    r.cov_stage += 1;
  }
  return gicp_run_start_outer(c, r);
}

int gicp_run_begin(icpgpu_ctx* c, GicpRun& r, int want_fitness, icpgpu_result* res, bool combine) {
  r = GicpRun{};
  r.combine = combine;
  r.t_start = std::chrono::steady_clock::now();
  r.res = res;
  r.want_fitness = want_fitness;
  {
    const int grc = ensure_gicp_resources(c);
    if (grc) return grc;
  }
  r.quadratic = gicp_inner_quadratic(c);
  if (r.quadratic) {
    const int qrc = ensure_gicp_quadratic_resources(c);
    if (qrc) return qrc;
  } else if (!gicp_run_device_ok(c)) {  // no device solver on this context: the blocking function
    r.phase = GicpRun::Blocking;
    return ICPGPU_OK;
  }
  init_result(res);
  c->prof.aligns += 1;
  c->prev.valid = c->tile_seed.valid = false;  // every alignment starts cold
  mat4f_identity(r.guess);
  mat4f_identity(r.transformation);
  mat4f_identity(r.previous);
  {
    int rc = resolve_sweep_timings(c, /*block=*/false);
    if (rc) return rc;
    c->dev_ms_accum = 0.0;
    c->call_sweeps = c->call_timed = 0;
  }
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  if (n_t == 0 || n_s < kGicpK || n_t < kGicpK) return gicp_run_finish_early(c, r);
  // GICP keeps d2 < r^2 (strict): the largest float below r^2
  const icpgpu_params& P = c->params;
  const double r2 = P.max_correspondence_distance * P.max_correspondence_distance;
  r.thr = threshold_from(r2);
  if ((double)r.thr >= r2) r.thr = std::nextafterf(r.thr, -INFINITY);
  r.thr_excl = std::nextafterf(r.thr, INFINITY);  // d2 < thr_excl  <=>  d2 <= thr
  r.cov_stage = 0;
  return gicp_run_next_cov(c, r);
}

static int gicp_run_restart_blocking(icpgpu_ctx* c, GicpRun& r) {
  c->prof.aligns -= 1;  // (align_gicp counts the alignment again)
  r.phase = GicpRun::Blocking;
  return ICPGPU_OK;
}

// the registration is over: result record, fitness sweep or Done
static int gicp_run_conclude(icpgpu_ctx* c, GicpRun& r) {
  float fin[16];
  gicp_compose_final(r.previous, r.guess, fin);
  std::memcpy(r.res->T, fin, sizeof(fin));
  for (int i = 0; i < 16; ++i) c->final_T[i] = (double)fin[i];
  c->have_final = true;
  r.res->converged = r.converged ? 1 : 0;
  r.res->iterations = r.nr;
  r.res->convergence_state = r.state;
  r.res->n_correspondences = r.n_corr;
  r.res->mse_last = r.mse;
  if (r.want_fitness) {
    r.phase = GicpRun::Fitness;
    r.polls = 0;
    r.t_issue = std::chrono::steady_clock::now();
    return sweep_issue(c, xform_from_f16(fin), FLT_MAX, /*open_range=*/true, r.ticket);
  }
  r.res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_start).count();
  r.phase = GicpRun::Done;
  return ICPGPU_OK;
}

// an outer iteration's minimiser is in: the new transformation, PCL's convergence test, then the end or the next outer iteration
static int gicp_run_after_solve(icpgpu_ctx* c, GicpRun& r, const Vec6& x) {
  const icpgpu_params& P = c->params;
  mat4f_identity(r.transformation);
  gicp_apply_state(r.transformation, x);
  const double delta = gicp_outer_delta(r.previous, r.transformation, 2e-3, P.transformation_epsilon);
  ++r.nr;
  c->prof.iterations += 1;
  if (r.nr >= P.max_iterations || (delta < 1 && !P.force_iterations)) {
    r.converged = true;
    r.state = r.nr >= P.max_iterations ? ICPGPU_CONV_ITERATIONS : ICPGPU_CONV_TRANSFORM;
    std::memcpy(r.previous, r.transformation, sizeof(r.previous));
    return gicp_run_conclude(c, r);
  }
  return gicp_run_queue_outer(c, r);
}

// One non-blocking step.  Returns < 0 on error, 0 when nothing has arrived yet, 1 when the run moved on (r.phase == Done: finished).
int gicp_run_step(icpgpu_ctx* c, GicpRun& r) {
  int rc;
  switch (r.phase) {
    case GicpRun::WantSolve:
      return 0;  // (the scheduler's move: gicp_run_solver_launched)
    case GicpRun::Blocking:
      if ((rc = align_gicp(c, nullptr, nullptr, r.want_fitness, r.res))) return rc;
      r.phase = GicpRun::Done;
      return 1;
    case GicpRun::CovGrid: {
      if (!run_marker_seen(c, r.marker)) {
        if ((++r.polls & 0x3FFu) != 0) return 0;
        const hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for an index build: %s", hipGetErrorString(q));
        if (q == hipErrorNotReady) {
          if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_issue).count() > wait_timeout_ms())
            return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for an index build (hung kernel?)", wait_timeout_ms());
          return 0;
        }
        // (the stream has drained: the read-back is there whether or not the marker's pair shows -- the stream's word is as good)
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if ((rc = gb_advance(c, r.gb))) return rc;
      const bool tgt = r.cov_stage == 0;
      GridIndex& G = tgt ? c->cov_grid_tgt : c->cov_grid_src;
      if (r.gb.state != GridBuild::Done) {
        r.marker = run_post_marker(c, static_cast<const int*>(G.ints.ptr), &rc);
        if (rc) return rc;
        r.polls = 0;
        return 1;
      }
      const Cloud& cloud = tgt ? c->tgt : c->src;
      if ((rc = cov_launch(c, cloud, tgt ? c->tgt_version : c->src_version, G, tgt ? c->cov_tgt : c->cov_src,
                           tgt ? c->cov_tgt_version : c->cov_src_version, /*timed=*/false)))
        return rc;
      r.cov_stage += 1;
      if ((rc = gicp_run_next_cov(c, r))) return rc;
      return 1;
    }
    case GicpRun::Solve: {
      double out[24];
      const int n = gicp_solve_out_granules();
      bool all = true;
      for (int k = 0; k < n; ++k) all = gicp_granule_read(c->h_solve + 2 * k, r.seq0, &out[k]) && all;
      bool gave_up = false;
      if (!all) {
        if ((++r.polls & 0x3FFu) != 0) return 0;
        const hipError_t q = hipStreamQuery(r.solve_stream ? r.solve_stream : c->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for the GICP device solver: %s", hipGetErrorString(q));
        if (q == hipErrorNotReady) {
          if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_issue).count() > wait_timeout_ms())
            return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for the GICP device solver (hung kernel?)", wait_timeout_ms());
          return 0;
        }
        all = true;
        for (int k = 0; k < n; ++k) all = gicp_granule_read(c->h_solve + 2 * k, r.seq0, &out[k]) && all;
        gave_up = !all;  // the stream went idle without an answer
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      const int status = gave_up ? (int)gicp::kDeviceError : (int)out[0];
      if (status == gicp::kDeviceError) {  // a gather timed out (or the kernel never answered): this pair goes through the blocking path
        // Under a batch a timeout can be a scheduling accident (a run's workgroups waiting for CUs other launches hold), so ONE
        // does not switch the worker's device solver off for good, as it did until round 6: three in a row do.  Said once per
        // process on stderr either way -- the batch keeps its results (same bits) and loses speed, which nobody would notice otherwise.
        static std::atomic<bool> said{false};
        if (!said.exchange(true) || std::getenv("ICPGPU_DEBUG"))
          fprintf(stderr, "[icpgpu] GICP batch: the device solver%s gave no answer for a run; that pair is solved through the blocking path "
                          "(reported once; ICPGPU_DEBUG=1 reports every time)\n", r.local ? " (one XCD)" : "");
        if (++c->gicp_device_failures >= 3) {
          if (r.local) c->gicp_local_ok = false;
          else c->gicp_device_ok = false;
        }
        return gicp_run_restart_blocking(c, r) ? -1 : 1;
      }
      c->gicp_device_failures = 0;
      c->prof.gicp_device_solves += 1;
      r.res->gicp_solver = ICPGPU_GICP_SOLVER_DEVICE;
      const double m = out[7], evals = out[10];
#if defined(ICPGPU_DEV_SWITCHES)
      {  // development flavour, ICPGPU_BATCH_TRACE=1: where the device solver's microseconds go with several runs in flight
        static const bool trace = [] { const char* e = std::getenv("ICPGPU_BATCH_TRACE"); return e && std::atoi(e) != 0; }();
        if (trace) {
          static std::atomic<unsigned long long> n_runs{0}, n_evals{0}, ph[7];
          for (int k = 0; k < 6; ++k) ph[k].fetch_add((unsigned long long)(out[12 + k] * 100.0));
          ph[6].fetch_add((unsigned long long)(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - r.t_issue).count() * 100.0));
          n_evals.fetch_add((unsigned long long)evals);
          if ((n_runs.fetch_add(1) & 1023u) == 1023u) {
            const double ne = (double)n_evals.load();
            fprintf(stderr, "[icpgpu] device solver, mean per evaluation over %llu runs: state -> transform %.2f | accumulate %.2f | reduce + publish %.2f | gather %.2f | "
                            "gradient %.2f | kernel total %.2f us; host: launch -> result %.2f us\n", n_runs.load(), ph[0].load() * 0.01 / ne, ph[1].load() * 0.01 / ne,
                    ph[2].load() * 0.01 / ne, ph[3].load() * 0.01 / ne, ph[4].load() * 0.01 / ne, ph[5].load() * 0.01 / ne, ph[6].load() * 0.01 / ne);
          }
        }
      }
#endif
      r.mse = m > 0 ? out[8] / m : 0.0;
      r.n_corr = (unsigned)m;
      c->prof.gicp_cost_launches += (uint64_t)evals;
      c->prof.gicp_eval_corr += (uint64_t)(m * evals);
      c->prof.gicp_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_issue).count();
      std::memcpy(r.previous, r.transformation, sizeof(r.previous));
      if (status == gicp::kNotEnoughPoints) {  // NotEnoughPointsException -> the loop breaks with converged_ = false
        r.state = ICPGPU_CONV_NO_CORRESPONDENCES;
        if ((rc = gicp_run_conclude(c, r))) return rc;
        return 1;
      }
      if (status != gicp::kOk) {  // SolverDidntConvergeException
        r.state = ICPGPU_NOT_CONVERGED;
        if ((rc = gicp_run_conclude(c, r))) return rc;
        return 1;
      }
      Vec6 x;
      for (int k = 0; k < 6; ++k) x[k] = out[1 + k];
      if ((rc = gicp_run_after_solve(c, r, x))) return rc;
      return 1;
    }
    case GicpRun::Quad: {
      double sums[2 * kGicpQuadSums];
      if (!quad_sums_read(c, r.seq0, sums)) {
        if ((++r.polls & 0x3FFu) != 0) return 0;
        const hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for the GICP quadratic pass: %s", hipGetErrorString(q));
        if (q == hipErrorNotReady) {
          if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_issue).count() > wait_timeout_ms())
            return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for the GICP quadratic pass (hung kernel?)", wait_timeout_ms());
          return 0;
        }
        if (!quad_sums_read(c, r.seq0, sums)) return fail(c, ICPGPU_ERR_HIP, "the GICP quadratic pass finished without publishing its sums");
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      c->quad_pending = false;
      const double m = sums[2 * 73], d2 = sums[2 * 74];
      r.mse = m > 0 ? d2 / m : 0.0;
      r.n_corr = (unsigned)m;
      std::memcpy(r.previous, r.transformation, sizeof(r.previous));
      if (r.n_corr < 4) {  // NotEnoughPointsException -> the loop breaks with converged_ = false
        r.state = ICPGPU_CONV_NO_CORRESPONDENCES;
        if ((rc = gicp_run_conclude(c, r))) return rc;
        return 1;
      }
      Vec6 x = gicp_state_from_matrix(r.transformation);
      int evals = 0;
      const GicpSolve sr = gicp_minimize_quadratic(sums, r.guess, x, 20, 1e-2, &evals);
      c->prof.gicp_quadratic_solves += 1;
      r.res->gicp_solver = ICPGPU_GICP_SOLVER_QUADRATIC;
      c->prof.gicp_cost_launches += (uint64_t)evals;
      c->prof.gicp_eval_corr += (uint64_t)(m * evals);
      c->prof.gicp_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_issue).count();
      if (sr != GicpSolve::Ok) {  // SolverDidntConvergeException
        r.state = ICPGPU_NOT_CONVERGED;
        if ((rc = gicp_run_conclude(c, r))) return rc;
        return 1;
      }
      if ((rc = gicp_run_after_solve(c, r, x))) return rc;
      return 1;
    }
    case GicpRun::Fitness: {
      if (!sweep_ready(c, r.ticket)) {
        if ((++r.polls & 0x3FFu) == 0) {
          const hipError_t q = hipStreamQuery(c->stream);
          if (q != hipSuccess && q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for a reduction: %s", hipGetErrorString(q));
          if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_issue).count() > wait_timeout_ms())
            return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for a kernel's result (hung kernel?)", wait_timeout_ms());
        }
        return 0;
      }
      if ((rc = sweep_complete(c, r.ticket))) return rc;
      r.res->fitness = c->h_sums[0] > 0.0 ? c->h_sums[16] / c->h_sums[0] : DBL_MAX;
      if ((rc = resolve_sweep_timings(c, /*block=*/false))) return rc;
      r.res->t_device_ms = 0.0;  // (not sampled on this path)
      r.res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_start).count();
      r.phase = GicpRun::Done;
      return 1;
    }
    default:
      return 0;
  }
}

}  // namespace icpgpu_impl

extern "C" {

int icpgpu_gicp_covariances(icpgpu_ctx* c, int of_target, double* out6) {
  ENTER(c);
  Cloud& cl = of_target ? c->tgt : c->src;
  if (!cl.set) return fail(c, ICPGPU_ERR_NO_INPUT, "gicp_covariances: cloud not set");
  if (cl.n && !out6) return fail(c, ICPGPU_ERR_INVALID_ARG, "null output");
  if (cl.n < (size_t)kGicpK) return fail(c, ICPGPU_ERR_INVALID_ARG, "GICP needs at least %d points per cloud", kGicpK);
  int rc = of_target ? ensure_covariances(c, c->tgt, c->tgt_version, c->cov_grid_tgt, c->cov_tgt, c->cov_tgt_version)
                     : ensure_covariances(c, c->src, c->src_version, c->cov_grid_src, c->cov_src, c->cov_src_version);
  if (rc) return rc;
  const DeviceBuf& cov = of_target ? c->cov_tgt : c->cov_src;
  HIP_TRY(c, hipMemcpyAsync(out6, cov.ptr, cl.n * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return resolve_cov_timing(c);
}

int icpgpu_gicp_quadratic_sums(icpgpu_ctx* c, const float* T, double* sums150) {
  ENTER(c);
  if (!sums150) return fail(c, ICPGPU_ERR_INVALID_ARG, "null output");
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "gicp_quadratic_sums: clouds not set");
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  if (n_s < kGicpK || n_t < kGicpK) return fail(c, ICPGPU_ERR_INVALID_ARG, "GICP needs at least %d points per cloud", kGicpK);
  int rc;
  if ((rc = ensure_gicp_resources(c)) || (rc = ensure_gicp_quadratic_resources(c))) return rc;
  if ((rc = ensure_covariances(c, c->tgt, c->tgt_version, c->cov_grid_tgt, c->cov_tgt, c->cov_tgt_version))) return rc;
  if ((rc = ensure_covariances(c, c->src, c->src_version, c->cov_grid_src, c->cov_src, c->cov_src_version))) return rc;
  const double r2 = c->params.max_correspondence_distance * c->params.max_correspondence_distance;
  float thr = threshold_from(r2);
  if ((double)thr >= r2) thr = std::nextafterf(thr, -INFINITY);
  const float thr_excl = std::nextafterf(thr, INFINITY);
  if ((rc = ensure_grid(c, thr))) return rc;
  if ((rc = ensure(c, c->keys, (size_t)n_s * sizeof(unsigned long long)))) return rc;
  auto* keys = static_cast<unsigned long long*>(c->keys.ptr);
  float guess[16];
  if (T) std::memcpy(guess, T, sizeof(guess));
  else mat4f_identity(guess);
  const Xform Tq = xform_from_f16(guess);
  Rot3d R;
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) R.m[3 * r + cc] = (double)guess[cc * 4 + r];
  c->prev.valid = false;
  if (grid_ready(c)) {
    HIP_TRY(c, launch_nn_grid_search(c->src.data(), n_s, grid_flags(c->grid, false), Tq, static_cast<const float4*>(c->grid.sorted.ptr),
                                     static_cast<const int*>(c->grid.cell_start.ptr), c->grid.g, thr, keys, nullptr, nullptr, nullptr, c->stream));
  } else {
    if ((rc = nn_keys_brute(c, c->tgt.data(), n_t, Tq, keys))) return rc;
  }
  const unsigned long long seq = ++c->quad_seq;
  if ((rc = quad_pass_launch(c, n_s, keys, thr_excl, R, seq))) return rc;
  if ((rc = wait_quad_sums(c, seq, sums150))) return rc;
  c->quad_pending = false;
  return resolve_cov_timing(c);
}

int icpgpu_gicp_quadratic_eval(const double* sums150, const float* base16, const double* x6, double* f, double* g6) {
  if (!sums150 || !base16 || !x6 || !f || !g6) return ICPGPU_ERR_INVALID_ARG;
  icpgpu::Vec6 x;
  for (int k = 0; k < 6; ++k) x[k] = x6[k];
  icpgpu::GicpEval e;
  icpgpu::gicp_quadratic_eval(sums150, base16, x, e);
  *f = e.f;
  for (int k = 0; k < 6; ++k) g6[k] = e.g[k];
  return ICPGPU_OK;
}

}  // extern "C"
