// icpgpu_index.cpp -- the uniform grid over a cloud (replaces the per-scan FLANN kd-tree build of a1): resumable builds,
// the cell-size rules, the source in cell order, and the dispatch of the grid search (kernels: icp_grid.hip).
#include "icp_ctx.h"


namespace icpgpu_impl {

static double sparse_population() {  // ICPGPU_SPARSE_POP overrides (tuning experiments only)
  static const double v = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_SPARSE_POP"); return e ? std::atof(e) : kSparseCellPopulation; }();
  return v;
}

// cells per cutoff; 4.5 unless ICPGPU_GRID_DIV overrides it (tuning experiments only).  4 until the dense-population rule
// moved to 44: a raw 200k scan (population 47 at gate / 4) then paid a second count pass (table memset, count, scan, host
// round trip: ~80 us of a 0.25 ms build) for every scan; at gate / 4.5 it lands on ~36 at once.  (A hint from the previous
// cloud would save the same, but an alignment's cell size -- hence its summation order, hence its last bits -- would then
// depend on what the context did before.)
static double grid_divisor() {
  static const double d = [] {
    const char* v = ICPGPU_DEV_ENV("ICPGPU_GRID_DIV");
    const double x = v ? std::atof(v) : 0.0;
    return (x >= 1.0 && x <= 64.0) ? x : 4.5;
  }();
  return d;
}



int gb_issue_count(icpgpu_ctx* c, GridBuild& b);
static int gb_with_bbox(icpgpu_ctx* c, GridBuild& b, const int enc[6], bool cached);

// queue the bounding box (or find the grid already built)
int gb_begin(icpgpu_ctx* c, GridBuild& b, const Cloud& cloud, uint64_t version, double cut, bool adapt, GridIndex& G,
             const int* orig_index, double knn_population, double h_start) {
  const bool post = b.post;  // (set by build_grid before the call)
  b = GridBuild{};
  b.post = post;
  b.h_start = h_start;
  b.cloud = &cloud;
  b.version = version;
  b.cut = cut;
  b.adapt = adapt;
  b.G = &G;
  b.orig_index = orig_index;
  b.knn_population = knn_population;
  const float cutoff = (float)cut;
  if (G.built && G.version == version && G.cutoff == cutoff) return ICPGPU_OK;  // (state Done)
  G.built = true;
  G.usable = false;
  G.adopted = false;
  G.version = version;
  G.cutoff = cutoff;
  G.cut_d = cut;
  int rc = ensure(c, G.ints, (6 + kGridStatInts) * sizeof(int));
  if (rc) return rc;
  int* d_ints = static_cast<int*>(G.ints.ptr);
  b.t0 = std::chrono::steady_clock::now();
  if (version != 0 && cloud.bbox_version == version) return gb_with_bbox(c, b, cloud.bbox_enc, !cloud.bbox_exact);  // a containing box is known: no pass, no round trip
  HIP_TRY(c, launch_bbox(cloud.data(), (int)cloud.n, d_ints, c->stream));
  if (!b.post) HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_ints, 6 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  b.state = GridBuild::WaitBbox;
  return ICPGPU_OK;
}

// the bounding box has arrived (the caller synchronised the stream): size the table, queue the count pass
int gb_on_bbox(icpgpu_ctx* c, GridBuild& b) {
  if (b.version != 0) {  // remembered with the cloud: its next grid (another cell size, the other role after a promote) starts here
    std::memcpy(b.cloud->bbox_enc, c->h_ints, sizeof(b.cloud->bbox_enc));
    b.cloud->bbox_version = b.version;
    b.cloud->bbox_exact = true;
  }
  return gb_with_bbox(c, b, c->h_ints, false);
}

static int gb_with_bbox(icpgpu_ctx* c, GridBuild& b, const int enc[6], bool cached) {
  decode_bbox(enc, b.lo, b.hi);
  b.box_cached = cached;
  if (cached)  // a box handed over by the voxel filter holds float means of its points: room for their rounding (gb_on_count checks)
    for (int a = 0; a < 3; ++a) {
      const float pad = std::ldexp(std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])), -10);
      if (std::isfinite(pad)) {
        b.lo[a] -= pad;
        b.hi[a] += pad;
      }
    }
  if (!(b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2])) {  // no finite point
    b.state = GridBuild::Done;
    return ICPGPU_OK;
  }
  // Cell size: a quarter of the cutoff (cube radii 1, 2, 4, 5 cells for an unmatched point), grown until the dense
  // table fits.  If that leaves a typical point sharing its cell with more than kDenseCellPopulation others (a submap
  // of many scans), the cells shrink so that this population comes down to ~kTargetCellPopulation (populations of
  // surface samples scale with h^2): the octant stage then still certifies most points (their neighbour is closer than
  // h/2) and reads 4x fewer candidates.  Measured at 200k x 1M: 126 -> ~84 us per iteration.
  b.h = b.h_start > 0.0 ? std::min(std::max(b.h_start, b.cut / 64.0), 8.0 * b.cut / grid_divisor()) : b.cut / grid_divisor();
  b.attempt = 0;
  b.shrunk = false;
  return gb_issue_count(c, b);
}

int gb_issue_count(icpgpu_ctx* c, GridBuild& b) {
  GridIndex& G = *b.G;
  const int n_t = (int)b.cloud->n;
  int* d_ints = static_cast<int*>(G.ints.ptr);
  double& h = b.h;
  const float *lo = b.lo, *hi = b.hi;
  GridDesc& g = b.g;
  long long nx, ny, nz;
  for (;;) {
    nx = (long long)std::floor((hi[0] - lo[0]) / h) + 3;
    ny = (long long)std::floor((hi[1] - lo[1]) / h) + 3;
    nz = (long long)std::floor((hi[2] - lo[2]) / h) + 3;
    // at most 2^14 cells per axis: the float cell coordinates the search reasons with are then exact to 2^-9 of a
    // cell, well inside the 1/64 safety margin of its distance tests (icp_grid_device.h)
    if (nx * ny * nz <= kMaxGridCells && nx < (1 << 14) && ny < (1 << 14) && nz < (1 << 14)) break;
    h *= 1.15;
    if (!std::isfinite(h)) {
      b.state = GridBuild::Done;
      return ICPGPU_OK;
    }
  }
  g.h = (float)h;
  g.inv_h = 1.0f / g.h;
  g.ox = lo[0] - g.h;
  g.oy = lo[1] - g.h;
  g.oz = lo[2] - g.h;
  g.nx = (int)nx;
  g.ny = (int)ny;
  g.nz = (int)nz;
  if (nz <= ny && !ICPGPU_DEV_ENV("ICPGPU_Z_OUTER")) {
    g.sy = g.nx * g.nz;  // y outermost, z in the middle (the usual case: a scene much wider than it is tall)
    g.sz = g.nx;
  } else {
    g.sy = g.nx;
    g.sz = g.nx * g.ny;
  }
  g.r_max = (int)std::ceil(b.cut / ((double)g.h * (double)kGridSafety));
  if (g.r_max < 1) g.r_max = 1;
  if (!std::isfinite(g.ox) || !std::isfinite(g.oy) || !std::isfinite(g.oz) || !(g.inv_h > 0.f) || !std::isfinite(g.inv_h)) {
    b.state = GridBuild::Done;
    return ICPGPU_OK;
  }
  const long long ncells = nx * ny * nz;
  int rc;
  // The cell table is the one buffer of a pair whose size depends on the pair's CONTENT (its box and cell size).  The workers
  // of a batch are dealt different pairs in every call, so each of them kept meeting a table larger than any it had held --
  // a hipFree (which synchronises the device) + hipMalloc in the middle of a running batch, for a dozen calls in a row (traced
  // in round 5: 17-19 ms for a call with reallocations, 12 ms without).  Workers therefore size their tables by the largest
  // ANY worker of the batch has needed, with room to spare: one growth per worker, in the call after the first.
  size_t cells_alloc = (size_t)(ncells + 1);
  if (c->shared_table_cells) {
    size_t seen = c->shared_table_cells->load(std::memory_order_relaxed);
    while (cells_alloc > seen && !c->shared_table_cells->compare_exchange_weak(seen, cells_alloc, std::memory_order_relaxed)) {}
    const size_t floor_cells = std::max(seen, cells_alloc);
    cells_alloc = std::min<size_t>(floor_cells + floor_cells / 4, (size_t)kMaxGridCells + 1);
    cells_alloc = std::max(cells_alloc, (size_t)(ncells + 1));
  }
  if ((rc = ensure(c, G.cell_start, cells_alloc * sizeof(int)))) return rc;
  if ((rc = ensure(c, G.cell_of_point, (size_t)n_t * sizeof(int)))) return rc;
  if ((rc = ensure(c, G.rank, (size_t)n_t * sizeof(int)))) return rc;
  if ((rc = ensure(c, G.block_sums, (cells_alloc / kScanItems + 2) * sizeof(int)))) return rc;
  HIP_TRY(c, launch_grid_count(b.cloud->data(), n_t, g, static_cast<int*>(G.cell_of_point.ptr), static_cast<int*>(G.rank.ptr),
                               static_cast<int*>(G.cell_start.ptr), static_cast<int*>(G.block_sums.ptr), d_ints + 6, c->stream));
  if (!b.post) HIP_TRY(c, hipMemcpyAsync(c->h_ints + 6, d_ints + 6, kGridStatInts * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  b.state = GridBuild::WaitCount;
  return ICPGPU_OK;
}

// the occupancy statistics have arrived: another count pass with other cells, or the scan + scatter
int gb_on_count(icpgpu_ctx* c, GridBuild& b) {
  GridIndex& G = *b.G;
  const int n_t = (int)b.cloud->n;
  int* d_ints = static_cast<int*>(G.ints.ptr);
  double& h = b.h;
  const double cut = b.cut;
  unsigned long long sumsq = 0;
  std::memcpy(&sumsq, c->h_ints + 8, sizeof sumsq);
  const int binned = c->h_ints[10];
  if (b.box_cached && binned != n_t) {
    // the cached box does not contain every point (a mean that left its points' box through float rounding: thousands of
    // points in one voxel far from the origin) or the cloud holds non-finite points: the cloud's own bounding-box pass decides
    b.cloud->bbox_version = 0;
    b.box_cached = false;
    HIP_TRY(c, launch_bbox(b.cloud->data(), n_t, d_ints, c->stream));
    if (!b.post) HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_ints, 6 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    b.state = GridBuild::WaitBbox;
    return ICPGPU_OK;
  }
  const double pop = binned > 0 ? (double)sumsq / (double)binned : 0.0;
  const int attempt = b.attempt;
  bool again = false;
  if (attempt == 0 && b.adapt && pop > kDenseCellPopulation) {
    static const double target_pop = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_TARGET_POP"); return e ? std::atof(e) : kTargetCellPopulation; }();
    const double h_new = std::max(h * std::sqrt(target_pop / pop), cut / 16.0);
    if (h_new < 0.9 * h) {
      h = h_new;
      b.shrunk = true;
      again = true;
    }
  }
  // k-nearest-neighbour searches (GICP covariances) read whole 3x3x3 cubes: they want cells for their own population (ensure_covariances)
  if (!again && attempt == 0 && b.knn_population > 0.0 && binned > 0 && (pop > 2.0 * b.knn_population || pop < 0.5 * b.knn_population)) {
    // ... in both directions: over a sparse (voxel-filtered) cloud the 20 neighbours lie ~2.5 point spacings away, and
    // cells that small make the search restart with cubes of 25 and 81 rows (22k-point cloud: 0.60 -> 0.35 ms)
    const double h_new = std::min(std::max(h * std::sqrt(b.knn_population / pop), cut / 64.0), 8.0 * h);
    if (h_new < 0.9 * h || h_new > 1.1 * h) {
      h = h_new;
      b.shrunk = true;
      again = true;
    }
  }
  // ... and a sparse target (one point per 0.5 m voxel: the mapper's nn cloud, a voxel-filtered scan) gets cells of
  // twice the size: neighbours are then typically farther than h/2 and would fall through the octant stage
  if (!again && attempt <= 1 && !b.shrunk && b.adapt && binned > 0 && pop < sparse_population() && 2.0 * h <= cut) {
    h *= 2.0;
    again = true;
  }
  if (again) {
    b.attempt += 1;
    return gb_issue_count(c, b);
  }
  G.n_binned = binned;
  G.max_pop = c->h_ints[7];
  G.point_population = pop;
  if (std::getenv("ICPGPU_DEBUG")) fprintf(stderr, "[icpgpu] grid n=%d binned=%d h=%.4f dims=%dx%dx%d pop=%.1f max=%d attempt=%d\n", n_t, binned, h, b.g.nx, b.g.ny, b.g.nz, pop, G.max_pop, attempt);
  int rc;
  if ((rc = ensure(c, G.sorted, (size_t)n_t * sizeof(float4)))) return rc;
  HIP_TRY(c, launch_grid_finish(b.cloud->data(), n_t, b.g, static_cast<const int*>(G.cell_of_point.ptr),
                                static_cast<const int*>(G.rank.ptr), static_cast<int*>(G.cell_start.ptr),
                                static_cast<int*>(G.block_sums.ptr), d_ints + 6, b.orig_index, static_cast<float4*>(G.sorted.ptr), c->stream));
  // No synchronisation here: the search that follows is queued behind the scan and the scatter.  (The build used to end
  // with a stream synchronisation only to time itself with events: 10-20 us of a 0.12 ms build.)  grid_build_ms is the
  // host's time in this function -- it contains the two round trips above, not the tail of the last three kernels.
  c->prof.grid_builds += 1;
  c->prof.grid_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b.t0).count();
  G.g = b.g;
  G.usable = G.n_binned > 0 && G.max_pop <= kMaxCellPopulation;
  static std::atomic<uint64_t> g_grid_serial{0};
  G.serial = ++g_grid_serial;
  b.state = GridBuild::Done;
  return ICPGPU_OK;
}

// A build whose count pass is queued (state WaitCount), finished WITHOUT its statistics: the cells stay as they were chosen, every
// point is ASSUMED binned.  For callers that know a containing box and a proven cell size (ensure_covariances with the box the voxel
// filter handed over and the last cloud's cell size) and check the statistics -- still in G.ints behind the count pass -- when
// they next wait for the device: one host round trip less per cloud.  The neighbours a search over this grid returns are exact
// whatever the cells are; only `every point binned` is a matter of correctness, and that is what the later check is for.
int gb_finish_unchecked(icpgpu_ctx* c, GridBuild& b) {
  if (b.state != GridBuild::WaitCount) return fail(c, ICPGPU_ERR_INVALID_ARG, "gb_finish_unchecked: no count pass in flight");
  GridIndex& G = *b.G;
  const int n_t = (int)b.cloud->n;
  int* d_ints = static_cast<int*>(G.ints.ptr);
  G.n_binned = n_t;
  G.max_pop = 0;
  G.point_population = b.knn_population > 0.0 ? b.knn_population : kTargetCellPopulation;
  int rc;
  if ((rc = ensure(c, G.sorted, (size_t)n_t * sizeof(float4)))) return rc;
  HIP_TRY(c, launch_grid_finish(b.cloud->data(), n_t, b.g, static_cast<const int*>(G.cell_of_point.ptr),
                                static_cast<const int*>(G.rank.ptr), static_cast<int*>(G.cell_start.ptr),
                                static_cast<int*>(G.block_sums.ptr), d_ints + 6, b.orig_index, static_cast<float4*>(G.sorted.ptr), c->stream));
  c->prof.grid_builds += 1;
  c->prof.grid_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b.t0).count();
  G.g = b.g;
  G.usable = n_t > 0;
  static std::atomic<uint64_t> g_spec_serial{1ull << 62};  // (its own range: never equal to a checked build's serial)
  G.serial = ++g_spec_serial;
  b.state = GridBuild::Done;
  return ICPGPU_OK;
}

// after the caller has synchronised the build's stream
int gb_advance(icpgpu_ctx* c, GridBuild& b) {
  if (b.state == GridBuild::WaitBbox) return gb_on_bbox(c, b);
  if (b.state == GridBuild::WaitCount) return gb_on_count(c, b);
  return ICPGPU_OK;
}

int build_grid(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, double cut, bool adapt, GridIndex& G,
               const int* orig_index, double knn_population, double h_start) {
  GridBuild b;
  b.post = true;
  int rc = gb_begin(c, b, cloud, version, cut, adapt, G, orig_index, knn_population, h_start);
  while (!rc && b.state != GridBuild::Done) {
    // the six box words or the occupancy statistics: a posted kernel + a polled mailbox (10 us) instead of a copy + a stream
    // synchronisation (16 us); gb_on_bbox / gb_on_count read them from c->h_ints as before
    const int* d_ints = static_cast<const int*>(G.ints.ptr);
    if (b.state == GridBuild::WaitBbox) rc = fetch_ints(c, d_ints, 6, c->h_ints);
    else rc = fetch_ints(c, d_ints + 6, kGridStatInts, c->h_ints + 6);
    if (!rc) rc = gb_advance(c, b);
  }
  return rc;
}

// The target's grid, if the parameters ask for it; otherwise (or when it cannot help) the brute-force kernel is used.
int ensure_grid(icpgpu_ctx* c, float accept_thr) {
  GridIndex& G = c->grid;
  const int mode = c->params.nn_mode;
  const bool want = mode == ICPGPU_NN_GRID || (mode == ICPGPU_NN_AUTO && c->tgt.n >= kGridMinTarget);
  const double cut = std::sqrt((double)accept_thr) * (1.0 + 1e-6);
  if (!want || c->tgt.n == 0 || !(accept_thr > 0.f) || !std::isfinite(cut) || cut > 1e6) {
    G.usable = false;
    G.built = false;
    return ICPGPU_OK;
  }
  // GICP: the grid the target's covariances were computed over (cells for a population of ~32: ensure_covariances) serves the
  // correspondence search as well.  The keys are exact whatever the cells are and nothing downstream of them depends on the
  // cell size, so the results are the same bits -- and the per-scan pipeline builds ONE index per cloud instead of two (the
  // second one cost 0.15 ms per scan: a bounding box and up to three count passes, each a host round trip).  The grids change
  // hands (a swap: no buffer is freed); other methods never see an adopted grid.
  static const bool adopt_enabled = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_GICP_ADOPT_GRID"); return !e || std::atoi(e) != 0; }();
  const bool gicp = c->params.method == ICPGPU_GICP && adopt_enabled;
  if (G.adopted) {
    if (gicp && G.built && G.usable && G.version == c->tgt_version && (double)accept_thr <= G.cut_d * G.cut_d) return ICPGPU_OK;
    G.built = G.usable = G.adopted = false;  // another method, another target or a wider gate: its own grid
  }
  if (gicp && !(G.built && G.version == c->tgt_version && G.cutoff == (float)cut)) {
    GridIndex& V = c->cov_grid_tgt;
    if (std::getenv("ICPGPU_DEBUG"))
      fprintf(stderr, "[icpgpu] ensure_grid (GICP): covariance grid built %d usable %d version %llu (target %llu, covariances %llu) cut %.3f thr %.6f\n",
              (int)V.built, (int)V.usable, (unsigned long long)V.version, (unsigned long long)c->tgt_version, (unsigned long long)c->cov_tgt_version, V.cut_d, (double)accept_thr);
    if (V.built && V.usable && V.version == c->tgt_version && c->cov_tgt_version == c->tgt_version && c->cov_tgt.ptr &&
        (double)accept_thr <= V.cut_d * V.cut_d) {
      std::swap(G, V);
      G.adopted = true;
      V.built = V.usable = V.adopted = false;
      c->prof.grid_adopted += 1;
      return ICPGPU_OK;
    }
  }
  return build_grid(c, c->tgt, c->tgt_version, cut, /*adapt=*/true, G);
}

constexpr double kPackRowsBelowPopulation = 30.0;  // see sweep_rows_packed

int grid_flags(const GridIndex& G, bool src_in_cell_order) {
  return (src_in_cell_order ? kGridSrcInCellOrder : 0) | (G.point_population < kPackRowsBelowPopulation ? kGridPackShortRows : 0) |
         (G.n_binned >= (1 << 28) ? kGridOver4GiB : 0);
}

// The previous-neighbour buffer for a sweep of src_pts[0..n_q) over G, or nullptr when the kernel chosen for this size
// keeps none (ICPGPU_PREV=0 switches the mechanism off: A/B measurements).  use = the entries come from a sweep of the
// same queries over the same target points.
int prev_neighbours(icpgpu_ctx* c, const GridIndex& G, const float4* src_pts, int n_q, int flags, unsigned int*& buf, bool& use) {
  static const bool enabled = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_PREV"); return !e || std::atoi(e) != 0; }();
  buf = nullptr;
  use = false;
  if (!enabled || !grid_search_keeps_prev(n_q, flags)) return ICPGPU_OK;
  PrevNeighbours& P = c->prev;
  const void* before = P.buf.ptr;
  int rc = ensure(c, P.buf, (size_t)n_q * sizeof(unsigned int));
  if (rc) return rc;
  use = P.valid && P.buf.ptr == before && P.src == src_pts && P.n == n_q && P.sorted == G.sorted.ptr &&
        P.grid_version == G.version && P.n_binned == G.n_binned && P.src_version == c->src_version;
  c->prof.grid_bounded += use ? 1 : 0;
  P.valid = true;
  P.src = src_pts;
  P.sorted = G.sorted.ptr;
  P.n = n_q;
  P.grid_version = G.version;
  P.n_binned = G.n_binned;  // the entries are positions in the sorted copy: ANY of its points bounds a search, whichever build
                            // of the grid put it there (a rebuild over the same target holds the same points, maybe elsewhere)
  P.src_version = c->src_version;
  buf = static_cast<unsigned int*>(P.buf.ptr);
  return ICPGPU_OK;
}

// Exact NN keys for every source point via the grid: points the grid cannot match within its cutoff are finished by
// the brute-force kernel. Does not synchronise except for the 4-byte unmatched count.
// With `deferred` the completion runs without a host round trip: the few-queries kernel reads the number of unmatched
// points on the device and leaves it in *deferred (mapped host memory); the caller looks at it once its own results have
// arrived and calls complete_deferred_keys() in the rare case that there were too many for that kernel.
int nn_keys_grid(icpgpu_ctx* c, GridIndex& G, const float4* src_pts, int n_s, const float4* tgt_pts, int n_t, const Xform& T,
                 unsigned long long* keys, int* deferred) {
  int rc = ensure(c, G.unmatched, (size_t)(n_s + 1) * sizeof(int));
  if (rc) return rc;
  int* d_list = static_cast<int*>(G.unmatched.ptr);
  int* d_count = d_list + n_s;
  HIP_TRY(c, hipMemsetAsync(d_count, 0, sizeof(int), c->stream));
  // A search WITHOUT a gate (getFitnessScore, icpgpu_nn, the map's nn cloud): the 1-2 % of the queries whose neighbour is
  // beyond the gate the grid was built for would all go to the brute-force completion (62 us at 50k x 50k, more than the
  // whole grid sweep).  Letting the cubes grow to 4x the gate settles nearly all of them in the grid: a few thousand
  // cell rows for a few hundred queries.
  GridDesc g_open = G.g;
  g_open.r_max = std::min(4 * G.g.r_max, 48);
  unsigned int* prev = nullptr;
  bool use_prev = false;
  if ((rc = prev_neighbours(c, G, src_pts, n_s, grid_flags(G, false), prev, use_prev))) return rc;
  HIP_TRY(c, launch_nn_grid_search(src_pts, n_s, grid_flags(G, false), T, static_cast<const float4*>(G.sorted.ptr),
                                   static_cast<const int*>(G.cell_start.ptr), g_open, 0.f, keys, nullptr, d_list, d_count,
                                   c->stream, prev, use_prev));
  if (deferred) {
    HIP_TRY(c, launch_nn_brute_few(src_pts, d_list, d_count, 0, tgt_pts, n_t, T, keys, deferred, c->stream));
    return ICPGPU_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int n_un = c->h_ints[0];
  c->prof.grid_fallback_points += (uint64_t)n_un;
  if (n_un > 0)
    HIP_TRY(c, launch_nn_brute_list(src_pts, d_list, n_un, tgt_pts, n_t, T, c->num_cus, keys, c->stream));
  return ICPGPU_OK;
}

// the tiled brute-force completion for a deferred search that listed more than kFewQueries points
int complete_deferred_keys(icpgpu_ctx* c, GridIndex& G, const float4* src_pts, int n_s, const float4* tgt_pts, int n_t,
                           const Xform& T, unsigned long long* keys, int n_un) {
  const int* d_list = static_cast<const int*>(G.unmatched.ptr);
  HIP_TRY(c, launch_nn_brute_list(src_pts, d_list, n_un, tgt_pts, n_t, T, c->num_cus, keys, c->stream));
  return ICPGPU_OK;
}

int nn_keys_grid(icpgpu_ctx* c, const Xform& T, unsigned long long* keys) {
  return nn_keys_grid(c, c->grid, c->src.data(), (int)c->src.n, c->tgt.data(), (int)c->tgt.n, T, keys);
}

bool grid_ready(const icpgpu_ctx* c) { return c->grid.usable && c->grid.version == c->tgt_version; }

// The grid search is bound by L2 traffic (TCC ~75 % busy at 200k x 200k): waves that run together should look at the
// same cells.  Binning the SOURCE with the same machinery gives a cell-ordered copy of it (measured 92 -> 75 us per
// iteration at 200k x 200k).  In the odometry loop this costs nothing: the cloud becomes the next target on
// promote_source_to_target and brings this grid along, so every cloud is binned exactly once.  The order of the source
// is irrelevant to the fused reduction; paths that return per-point results keep the caller's order.
int source_order_mode() {  // ICPGPU_ORDER_SOURCE=0/1 overrides the size rule (experiments)
  static const int m = [] { const char* v = ICPGPU_DEV_ENV("ICPGPU_ORDER_SOURCE"); return v ? std::atoi(v) : -1; }();
  return m;
}

int ensure_source_order(icpgpu_ctx* c, float accept_thr) {
  GridIndex& G = c->src_grid;
  const int mode = source_order_mode();
  const bool want = grid_ready(c) && (mode == 1 || (mode != 0 && c->src.n >= kOrderSourceMin));
  if (!want) {
    if (G.version != c->src_version) G.built = G.usable = false;
    return ICPGPU_OK;
  }
  return build_grid(c, c->src, c->src_version, std::sqrt((double)accept_thr) * (1.0 + 1e-6), /*adapt=*/true, G);
}

bool source_ordered(const icpgpu_ctx* c) {
  return c->src_grid.built && c->src_grid.usable && c->src_grid.version == c->src_version && c->src_grid.n_binned > 0;
}

// Brute-force keys of every source point against tgt_pts (exact NN).  Large problems go to the matrix
// cores (icp_brute_mfma.hip: an MFMA lower bound settles all but a handful of pairs, those are evaluated exactly); that
// kernel wants its sources as neighbours in space, so the source is binned with the grid machinery first (cached per source
// cloud; promote_source_to_target hands the same structure on as the next target's grid) and then put in Morton order of its
// cells.  brute_variant: 0 = this choice (the bound on the bf16 matrix path, icp_brute_bf16.hip), 2 = the bound in f32 MFMAs
// (icp_brute_mfma.hip, round 2's kernel), 1 = the plain-VALU kernel whatever the size (A/B measurements; ICPGPU_NN_VARIANT
// picks among its variants).  All three return the same keys, bit for bit.

// c->brute_order.pts = the source's cell-ordered copy (c->src_grid, built and usable) in Morton order of its cells
int source_in_morton_order(icpgpu_ctx* c) {
  BruteOrder& O = c->brute_order;
  const int n_b = c->src_grid.n_binned;
  if (O.valid && O.grid_serial == c->src_grid.serial && O.n == n_b) return ICPGPU_OK;
  int rc;
  if ((rc = ensure(c, O.pts, (size_t)n_b * sizeof(float4)))) return rc;
  if ((rc = ensure(c, O.work, morton_order_work_ints(n_b) * sizeof(int)))) return rc;
  HIP_TRY(c, launch_morton_order(static_cast<const float4*>(c->src_grid.sorted.ptr), n_b, c->src_grid.g,
                                 static_cast<int*>(O.work.ptr), static_cast<float4*>(O.pts.ptr), c->stream));
  O.valid = true;
  O.grid_serial = c->src_grid.serial;
  O.n = n_b;
  return ICPGPU_OK;
}

}  // namespace icpgpu_impl
