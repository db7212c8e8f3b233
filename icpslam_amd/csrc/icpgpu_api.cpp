// icpgpu_api.cpp -- the C-ABI of libicpgpu.so (include/icpgpu.h): context, buffers, the ICP iteration loop.
//
// Host side of the hot path behind the PCL Registration protocol used at
//   /root/reference/src/icpslam/icp_odometer.cpp:188-201 and src/icpslam/octree_mapper.cpp:104-117.
// Per iteration: [fill keys] -> NN kernel (a2) -> reduce kernels (a3+a4) -> 136-byte D2H -> host SVD (a5) + convergence
// test (a7).  The source cloud is never rewritten: every kernel applies the accumulated transform on load (a6 fused).
// There is deliberately no CPU fallback: if HIP or the device is unusable every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#if defined(__linux__)
#include <sched.h>
#endif
#include <atomic>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <string>
#include <thread>
#include <vector>

#include "../../include/icpgpu.h"
#include "icp_kernels.h"
#include "icp_gicp_solver.h"
#include "icp_solver.h"

using namespace icpgpu;

namespace {

thread_local std::string g_create_error;

struct DeviceBuf {
  void* ptr = nullptr;
  size_t cap = 0;      // bytes owned (0 when external)
  bool external = false;
};

struct Cloud {
  DeviceBuf buf;
  size_t n = 0;
  bool set = false;
  // fingerprint of 256 sampled points, taken from the HOST buffer at upload: lets icpgpu_set_target dismiss a same-sized but
  // different cloud (fixed-size scans!) in a microsecond instead of hashing megabytes to find out
  unsigned long long sample_fp = 0;
  bool sample_valid = false;
  const float4* data() const { return static_cast<const float4*>(buf.ptr); }
};

// Uniform grid over the target (icp_grid.hip), built lazily for the cutoff of the current parameters.
struct GridIndex {
  bool built = false, usable = false;
  uint64_t version = 0;      // target version it was built for
  float cutoff = 0.f;        // correspondence distance it was built for
  GridDesc g{};
  int n_binned = 0, max_pop = 0;
  double point_population = 0.0;  // cell population seen by a random point (sum c^2 / sum c)
  DeviceBuf sorted, cell_start, cell_of_point, rank, block_sums, ints, unmatched, leftover;
  uint64_t serial = 0;       // unique per completed build (grids change hands between c->grid and c->src_grid)
};

// The neighbour every query found in the last grid sweep (nn_quad_kernel): an upper bound for the next sweep of the same
// queries over the same target, whatever the transform (icp_grid.hip).  `valid` is cleared at the start of every
// alignment, so an alignment never depends on the one before it.
struct PrevNeighbours {
  DeviceBuf buf;
  bool valid = false;
  const void* src = nullptr;     // query array the entries are indexed by
  const void* sorted = nullptr;  // grid they were found in
  int n = 0;
  uint64_t grid_version = 0, src_version = 0;
};

// Keys of the last matrix-core brute-force sweep (icp_brute_mfma.hip): every source's neighbour, a bound for the next sweep
// of the same source over the same target.
// The source in Morton order of its grid cells (icp_brute_bf16.hip wants a workgroup's sources close together): cached per
// build of the source's grid.
struct BruteOrder {
  DeviceBuf pts, work, check;
  bool valid = false;
  uint64_t grid_serial = 0;
  int n = 0;
};

// Keys of the last tile-search sweep (icp_tile.hip) of this alignment: every source's neighbour, a radius for the next sweep.
struct TileSeed {
  DeviceBuf keys, stats, prev;
  bool valid = false;
  uint64_t src_version = 0, grid_version = 0;
  int n_s = 0;
};

struct BruteSeed {
  DeviceBuf keys;
  bool valid = false;
  uint64_t src_version = 0, tgt_version = 0;
  const void* tgt = nullptr;
  int n_s = 0, n_t = 0;
};

// The mapper's one-point-per-voxel map (icp_map.hip; octree_mapper.cpp:55-90).
struct VoxelMap {
  bool defined = false;   // resolution set by icpgpu_map_reset
  bool anchored = false;  // lattice origin fixed by the first point ever added
  MapDesc desc{};
  int n = 0;              // points in the map
  uint64_t version = 1;   // bumped whenever points are appended
  unsigned int cap = 0;   // hash-set capacity (power of two, load <= 1/2)
  Cloud pts;              // the map cloud (owned)
  DeviceBuf keys, vals, first, staged, moved, slot_of, flags, rank, temp, counter, nn_keys;
  DeviceBuf first_user, uflags, urank, uniq_index;
  Cloud uniq;             // the distinct points of the last nn cloud (what the ICP target's grid is built from)
  GridIndex grid;         // for the nn-cloud search
  // PCL-faithful approxNearestSearch mode (icpgpu_map_set_search): the octree's bounding box as PCL grows it, replayed over
  // the map points in insertion order, and the hash set of occupied octree nodes (icp_map.hip)
  int search_mode = ICPGPU_MAP_SEARCH_EXACT;
  bool box_defined = false;
  ApproxBox box{};
  int box_upto = 0;           // map points already folded into the box
  uint64_t box_version = 0;   // bumped whenever the box grows
  DeviceBuf node_keys, node_vals;
  unsigned int node_cap = 0;
  int nodes_upto = 0;         // map points whose paths are in the node set
  uint64_t nodes_box_version = ~0ull;
};

constexpr long long kMaxGridCells = 16ll << 20; // 64 MB of cell_start at most
constexpr int kMaxCellPopulation = 4096;        // beyond this a lane's serial cell scan is slower than brute force
constexpr double kDenseCellPopulation = 44.0;   // shrink the cells beyond this point-weighted population (64 until round 2: a 200k scan at gate / 4 sits at 47; with cells for ~36 its alignment takes 620-645 instead of 668 us of search) ...
constexpr double kTargetCellPopulation = 36.0;  // ... down to about this one (18 until the search was ball-pruned: 200k x 1M from identity 77 -> 65 us)
constexpr double kSparseCellPopulation = 20.0;  // double the cells below this one (2.5 until then: 50k x 50k from identity 31 -> 27 us)
constexpr size_t kOrderSourceMin = 100000;      // AUTO: order the source by cell from this size on (see ensure_source_order)
constexpr int kEventRing = 64;                   // sweeps whose kernel timing may be outstanding
constexpr size_t kGridMinTarget = 4096;         // AUTO: below this the brute-force kernel is launch-latency bound anyway

}  // namespace

constexpr size_t kMaxServerWorkers = 8;

static bool gicp_server_enabled() {  // ICPGPU_GICP_SERVER=0: every GICP evaluation is its own launch
#if defined(__x86_64__)
  static const bool v = [] { const char* e = std::getenv("ICPGPU_GICP_SERVER"); return !e || std::atoi(e) != 0; }();
  return v;
#else
  return false;  // the command protocol relies on x86 store ordering
#endif
}

struct icpgpu_ctx {
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  icpgpu_params params{};
  Cloud src, tgt;
  DeviceBuf keys, partials, sums, out, idx, d2, cand_counter;
  GridIndex grid;            // acceleration structure over the current target
  GridIndex src_grid;        // the source in cell order (and, after promote_source_to_target, the next target's grid)
  PrevNeighbours prev;       // last sweep's neighbours (search bound of the next sweep)
  BruteSeed brute_seed;      // the same for the matrix-core brute-force kernels
  BruteOrder brute_order;    // the source in Morton order (bf16 matrix-core kernels)
  TileSeed tile_seed;        // the previous sweep's keys (matrix-core grid search)
  VoxelMap map;              // the mapper's map
  uint64_t tgt_version = 1;  // bumped whenever the target cloud changes
  // content fingerprints of the clouds in HBM (icpgpu_set_target's recognition of a cloud it already holds), computed on
  // demand and cached per version
  unsigned long long src_fp = 0, tgt_fp = 0;
  uint64_t src_fp_version = 0, tgt_fp_version = 0;
  DeviceBuf fp_acc;
  uint64_t src_version = 1;  // bumped whenever the source cloud changes
  // GICP mode: grids used for the 20-NN covariances, per-point covariances (6 doubles), Mahalanobis matrices
  GridIndex cov_grid_src, cov_grid_tgt;
  DeviceBuf cov_src, cov_tgt, maha;
  uint64_t cov_src_version = 0, cov_tgt_version = 0;
  // Per-iteration result mailbox in pinned, mapped host memory: 17 sums + 17 sequence flags.  The final reduction
  // stores straight into it and the host polls the flags -- no copy engine and no stream synchronisation (whose wake-up
  // costs 20-70 us depending on how the process set up the runtime) on the iteration path.
  double* h_sums = nullptr;
  double* h_sums_dev = nullptr;              // device alias of h_sums
  volatile unsigned long long* h_flags = nullptr;
  unsigned long long* h_flags_dev = nullptr;  // device alias of h_flags
  unsigned long long sums_seq = 0;
  // GICP cost evaluations: per-workgroup partials (kGicpDirectBlocks x 17) + one flag per workgroup, same kind of memory
  double* h_gicp = nullptr;
  double* h_gicp_dev = nullptr;
  volatile unsigned long long* h_gicp_flags = nullptr;
  unsigned long long* h_gicp_flags_dev = nullptr;
  // resident evaluation server (icp_gicp.hip): its command line, fine-grained device memory the host writes through the BAR
  unsigned int* gicp_cmd = nullptr;
  bool gicp_server_on = false;
  bool gicp_server_allowed = true;  // align_batch with more than kMaxServerWorkers threads: single launches (below)
  int gicp_blocks_most = kGicpDirectBlocks;
  // ICPGPU_GICP_TIMING=1 (development): where an evaluation's microseconds go, printed when the context is destroyed
  double gt_cmd = 0, gt_wait = 0, gt_merge = 0, gt_between = 0, gt_dev_wait = 0, gt_dev_work = 0;
  unsigned long long gt_n = 0;
  // ICPGPU_P2P_TIMING=1 (development): host time between a sweep's sums and the next search kernel's launch
  double pt_wait = 0, pt_solve = 0, pt_prelaunch = 0, pt_launch = 0, pt_rest = 0;
  unsigned long long pt_n = 0;
  std::chrono::steady_clock::time_point pt_ready{}, pt_issue_in{};
  std::chrono::steady_clock::time_point gt_last{};  // workgroups per cost evaluation: the whole chip, or this worker's share of it
  // kernel timing for the profile: event triples are recorded per sweep and only read back when the align ends
  std::vector<hipEvent_t> ev_ring;            // 3 * kEventRing events
  struct PendingSweep { int slot; bool grid; };
  std::vector<PendingSweep> pending;
  double dev_ms_accum = 0.0;
  unsigned call_sweeps = 0, call_timed = 0;  // sweeps of the current align call: all / timed
  int timing_every = 13;      // time one sweep in 13 (coprime with the 10 / 30 iterations of the reference's aligns; 7 until round 2: an event triple costs 6-7 us)
  unsigned sweep_counter = 0;
  int* h_ints = nullptr;     // pinned (16 ints: bbox / stats / counters)
  bool have_final = false;
  Mat4d final_T = mat4_identity();
  icpgpu_profile prof{};
  int nn_variant = -1;  // ICPGPU_NN_VARIANT: a variant of the plain-VALU brute-force kernel (-1: none forced)
  DeviceBuf vox_in, vox_out, vox_keys, vox_vals, vox_flags, vox_slots, vox_temp, vox_ints;  // voxel filter scratch
  DeviceBuf vox_bins, vox_pub;   // ... of the direct (no library sort) path: self-cleaning histogram + group ranges; published counts
  size_t vox_last_n = 0;         // points of the last icpgpu_voxel_grid result (still in vox_out)
  void* vox_bins_zeroed = nullptr;  // the allocation (address, size) whose histogram is known to be zero
  size_t vox_bins_zeroed_cap = 0;
  void* vox_pub_zeroed = nullptr;
  size_t vox_pub_zeroed_cap = 0;
  std::vector<icpgpu_ctx*> workers;  // align_batch: one sub-context (own stream + scratch) per host worker thread
  DeviceBuf batch_table;             // lock-step batch: the BatchPair table of the group this context leads
  int host_share = 1;                // batch drivers of this process that share its CPUs with this context (icp_multi.cpp)
  std::string err;
};

namespace {

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

#define HIP_TRY(c, expr)                                                                              \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail((c), e_ == hipErrorOutOfMemory ? ICPGPU_ERR_OOM : ICPGPU_ERR_HIP, "%s failed: %s", #expr, \
                  hipGetErrorString(e_));                                                             \
  } while (0)

int ensure(icpgpu_ctx* c, DeviceBuf& b, size_t bytes) {
  if (b.external) {
    b.ptr = nullptr;
    b.cap = 0;
    b.external = false;
  }
  if (bytes <= b.cap) return ICPGPU_OK;
  if (b.ptr) HIP_TRY(c, hipFree(b.ptr));
  b.ptr = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4;  // amortise growth across scans of slightly different size
  if (want < 256) want = 256;
  HIP_TRY(c, hipMalloc(&b.ptr, want));
  b.cap = want;
  return ICPGPU_OK;
}

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

int set_cloud_host(icpgpu_ctx* c, Cloud& cl, const float* xyzw, size_t n, bool sync = true) {
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

static double sparse_population() {  // ICPGPU_SPARSE_POP overrides (tuning experiments only)
  static const double v = [] { const char* e = std::getenv("ICPGPU_SPARSE_POP"); return e ? std::atof(e) : kSparseCellPopulation; }();
  return v;
}

// cells per cutoff; 4.5 unless ICPGPU_GRID_DIV overrides it (tuning experiments only).  4 until the dense-population rule
// moved to 44: a raw 200k scan (population 47 at gate / 4) then paid a second count pass (table memset, count, scan, host
// round trip: ~80 us of a 0.25 ms build) for every scan; at gate / 4.5 it lands on ~36 at once.  (A hint from the previous
// cloud would save the same, but an alignment's cell size -- hence its summation order, hence its last bits -- would then
// depend on what the context did before.)
static double grid_divisor() {
  static const double d = [] {
    const char* v = std::getenv("ICPGPU_GRID_DIV");
    const double x = v ? std::atof(v) : 0.0;
    return (x >= 1.0 && x <= 64.0) ? x : 4.5;
  }();
  return d;
}

// (Re)build a uniform grid over `cloud` for the cutoff `cut` (see icp_grid.hip).  G.usable stays false when the grid
// cannot help (no finite point, one cell holding > kMaxCellPopulation points).
//
// A build has two host round trips (the bounding box sizes the table; the occupancy statistics of the count pass may
// change the cell size once or twice).  It is written as a resumable state machine so that the lock-step batch path
// (icpgpu_align_batch) can take K builds through their round trips TOGETHER -- one stream synchronisation per stage for
// the whole group instead of two or three per pair; build_grid() drives one build to the end.
struct GridBuild {
  enum State { Done, WaitBbox, WaitCount };
  State state = Done;
  const Cloud* cloud = nullptr;
  uint64_t version = 0;
  double cut = 0.0;
  bool adapt = false;
  GridIndex* G = nullptr;
  const int* orig_index = nullptr;
  double knn_population = 0.0;
  double h = 0.0;
  int attempt = 0;
  bool shrunk = false;
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  GridDesc g{};
  std::chrono::steady_clock::time_point t0{};
};

int gb_issue_count(icpgpu_ctx* c, GridBuild& b);

// queue the bounding box (or find the grid already built)
int gb_begin(icpgpu_ctx* c, GridBuild& b, const Cloud& cloud, uint64_t version, double cut, bool adapt, GridIndex& G,
             const int* orig_index = nullptr, double knn_population = 0.0) {
  b = GridBuild{};
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
  G.version = version;
  G.cutoff = cutoff;
  int rc = ensure(c, G.ints, (6 + kGridStatInts) * sizeof(int));
  if (rc) return rc;
  int* d_ints = static_cast<int*>(G.ints.ptr);
  b.t0 = std::chrono::steady_clock::now();
  HIP_TRY(c, launch_bbox(cloud.data(), (int)cloud.n, d_ints, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_ints, 6 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  b.state = GridBuild::WaitBbox;
  return ICPGPU_OK;
}

// the bounding box has arrived (the caller synchronised the stream): size the table, queue the count pass
int gb_on_bbox(icpgpu_ctx* c, GridBuild& b) {
  decode_bbox(c->h_ints, b.lo, b.hi);
  if (!(b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2])) {  // no finite point
    b.state = GridBuild::Done;
    return ICPGPU_OK;
  }
  // Cell size: a quarter of the cutoff (cube radii 1, 2, 4, 5 cells for an unmatched point), grown until the dense
  // table fits.  If that leaves a typical point sharing its cell with more than kDenseCellPopulation others (a submap
  // of many scans), the cells shrink so that this population comes down to ~kTargetCellPopulation (populations of
  // surface samples scale with h^2): the octant stage then still certifies most points (their neighbour is closer than
  // h/2) and reads 4x fewer candidates.  Measured at 200k x 1M: 126 -> ~84 us per iteration.
  b.h = b.cut / grid_divisor();
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
  if (nz <= ny && !std::getenv("ICPGPU_Z_OUTER")) {
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
  const int nb = (int)((ncells + kScanItems - 1) / kScanItems);
  int rc;
  if ((rc = ensure(c, G.cell_start, (size_t)(ncells + 1) * sizeof(int)))) return rc;
  if ((rc = ensure(c, G.cell_of_point, (size_t)n_t * sizeof(int)))) return rc;
  if ((rc = ensure(c, G.rank, (size_t)n_t * sizeof(int)))) return rc;
  if ((rc = ensure(c, G.block_sums, (size_t)(nb + 1) * sizeof(int)))) return rc;
  HIP_TRY(c, launch_grid_count(b.cloud->data(), n_t, g, static_cast<int*>(G.cell_of_point.ptr), static_cast<int*>(G.rank.ptr),
                               static_cast<int*>(G.cell_start.ptr), static_cast<int*>(G.block_sums.ptr), d_ints + 6, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_ints + 6, d_ints + 6, kGridStatInts * sizeof(int), hipMemcpyDeviceToHost, c->stream));
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
  const double pop = binned > 0 ? (double)sumsq / (double)binned : 0.0;
  const int attempt = b.attempt;
  bool again = false;
  if (attempt == 0 && b.adapt && pop > kDenseCellPopulation) {
    static const double target_pop = [] { const char* e = std::getenv("ICPGPU_TARGET_POP"); return e ? std::atof(e) : kTargetCellPopulation; }();
    const double h_new = std::max(h * std::sqrt(target_pop / pop), cut / 16.0);
    if (h_new < 0.9 * h) {
      h = h_new;
      b.shrunk = true;
      again = true;
    }
  }
  // k-nearest-neighbour searches (GICP covariances) read whole 3x3x3 cubes: they want only a few points per cell
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

// after the caller has synchronised the build's stream
int gb_advance(icpgpu_ctx* c, GridBuild& b) {
  if (b.state == GridBuild::WaitBbox) return gb_on_bbox(c, b);
  if (b.state == GridBuild::WaitCount) return gb_on_count(c, b);
  return ICPGPU_OK;
}

int build_grid(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, double cut, bool adapt, GridIndex& G,
               const int* orig_index = nullptr, double knn_population = 0.0) {
  GridBuild b;
  int rc = gb_begin(c, b, cloud, version, cut, adapt, G, orig_index, knn_population);
  while (!rc && b.state != GridBuild::Done) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = gb_advance(c, b);
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
int prev_neighbours(icpgpu_ctx* c, const GridIndex& G, const float4* src_pts, int n_q, int flags, float4*& buf, bool& use) {
  static const bool enabled = [] { const char* e = std::getenv("ICPGPU_PREV"); return !e || std::atoi(e) != 0; }();
  buf = nullptr;
  use = false;
  if (!enabled || !grid_search_keeps_prev(n_q, flags)) return ICPGPU_OK;
  PrevNeighbours& P = c->prev;
  const void* before = P.buf.ptr;
  int rc = ensure(c, P.buf, (size_t)n_q * sizeof(float4));
  if (rc) return rc;
  use = P.valid && P.buf.ptr == before && P.src == src_pts && P.n == n_q && P.sorted == G.sorted.ptr &&
        P.grid_version == G.version && P.src_version == c->src_version;
  c->prof.grid_bounded += use ? 1 : 0;
  P.valid = true;
  P.src = src_pts;
  P.sorted = G.sorted.ptr;
  P.n = n_q;
  P.grid_version = G.version;
  P.src_version = c->src_version;
  buf = static_cast<float4*>(P.buf.ptr);
  return ICPGPU_OK;
}

// Exact NN keys for every source point via the grid: points the grid cannot match within its cutoff are finished by
// the brute-force kernel. Does not synchronise except for the 4-byte unmatched count.
// With `deferred` the completion runs without a host round trip: the few-queries kernel reads the number of unmatched
// points on the device and leaves it in *deferred (mapped host memory); the caller looks at it once its own results have
// arrived and calls complete_deferred_keys() in the rare case that there were too many for that kernel.
int nn_keys_grid(icpgpu_ctx* c, GridIndex& G, const float4* src_pts, int n_s, const float4* tgt_pts, int n_t, const Xform& T,
                 unsigned long long* keys, int* deferred = nullptr) {
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
  float4* prev = nullptr;
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
static int source_order_mode() {  // ICPGPU_ORDER_SOURCE=0/1 overrides the size rule (experiments)
  static const int m = [] { const char* v = std::getenv("ICPGPU_ORDER_SOURCE"); return v ? std::atoi(v) : -1; }();
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
constexpr int kMfmaMinPoints = 8192;

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

int nn_keys_brute(icpgpu_ctx* c, const float4* tgt_pts, int n_t, const Xform& T, unsigned long long* keys, bool* used_mfma = nullptr) {
  const int n_s = (int)c->src.n;
  if (used_mfma) *used_mfma = false;
  if (n_s <= 0) return ICPGPU_OK;
  if (c->params.brute_variant != 1 && c->nn_variant < 0 && n_s >= kMfmaMinPoints && n_t >= kMfmaMinPoints) {
    double cut = c->params.max_correspondence_distance;
    if (!(cut > 1e-3) || !std::isfinite(cut) || cut > 1e6) cut = 1.0;
    const float thr = threshold_from(cut * cut);
    int rc = build_grid(c, c->src, c->src_version, std::sqrt((double)thr) * (1.0 + 1e-6), /*adapt=*/true, c->src_grid);
    if (rc) return rc;
    if (c->src_grid.usable && c->src_grid.n_binned > 0) {
      // the keys of the last matrix-core sweep of this source over this target seed the next one (a private copy: c->keys
      // has other writers); results do not depend on the seed, only the time does
      BruteSeed& S = c->brute_seed;
      const bool seeded = S.valid && S.src_version == c->src_version && S.tgt_version == c->tgt_version && S.tgt == tgt_pts &&
                          S.n_s == n_s && S.n_t == n_t;
      if ((rc = ensure(c, S.keys, (size_t)n_s * sizeof(unsigned long long)))) return rc;
      HIP_TRY(c, launch_fill_keys(keys, n_s, c->stream));
      const unsigned long long* seed_keys = seeded ? static_cast<const unsigned long long*>(S.keys.ptr) : nullptr;
      if (c->params.brute_variant == 2) {
        HIP_TRY(c, launch_nn_brute_mfma(static_cast<const float4*>(c->src_grid.sorted.ptr), c->src_grid.n_binned, tgt_pts, n_t, T,
                                        c->num_cus, keys, seed_keys, c->stream));
      } else {
        BruteOrder& O = c->brute_order;
        const int n_b = c->src_grid.n_binned;
        if ((rc = source_in_morton_order(c))) return rc;
        // test mode (ICPGPU_MFMA_CHECK_BOUND=1, read per call): every pair evaluated exactly against its bound; the counters
        // land in the profile (brute_bound_violations must stay 0)
        const char* chk = getenv("ICPGPU_MFMA_CHECK_BOUND");
        unsigned long long* d_check = nullptr;
        if (chk && atoi(chk)) {
          if ((rc = ensure(c, O.check, 2 * sizeof(unsigned long long)))) return rc;
          d_check = static_cast<unsigned long long*>(O.check.ptr);
          HIP_TRY(c, hipMemsetAsync(d_check, 0, 2 * sizeof(unsigned long long), c->stream));
        }
        HIP_TRY(c, launch_nn_brute_bf16(static_cast<const float4*>(O.pts.ptr), n_b, tgt_pts, n_t, T, c->num_cus, keys, seed_keys,
                                        d_check, c->stream));
        if (d_check) {
          unsigned long long h[2] = {0, 0};
          HIP_TRY(c, hipMemcpyAsync(h, d_check, sizeof(h), hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(c, hipStreamSynchronize(c->stream));
          c->prof.brute_bound_violations += h[0];
          float worst;
          const unsigned int bits = (unsigned int)h[1];
          std::memcpy(&worst, &bits, 4);
          if ((double)worst > c->prof.brute_bound_worst) c->prof.brute_bound_worst = worst;
        }
      }
      HIP_TRY(c, hipMemcpyAsync(S.keys.ptr, keys, (size_t)n_s * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
      S.valid = true;
      S.src_version = c->src_version;
      S.tgt_version = c->tgt_version;
      S.tgt = tgt_pts;
      S.n_s = n_s;
      S.n_t = n_t;
      if (used_mfma) *used_mfma = true;
      return ICPGPU_OK;
    }
  }
  const NnPlan plan = plan_nn_brute(n_s, n_t, c->nn_variant < 0 ? 0 : c->nn_variant, c->num_cus);
  if (plan.splits > 1 && n_t > 0) HIP_TRY(c, launch_fill_keys(keys, n_s, c->stream));
  HIP_TRY(c, launch_nn_brute(c->src.data(), n_s, tgt_pts, n_t, T, plan, keys, c->stream));
  return ICPGPU_OK;
}

// Read back the kernel timings of the sweeps issued since the last call (one stream synchronisation for all of them).
// With block = false nothing waits: the events of the last sweep are normally complete a few microseconds after its
// result reached the mailbox (a short poll of hipEventQuery); if they are not, the timings stay pending and are read by
// the next call.  A stream synchronisation here costs 20-70 us of wake-up latency per alignment -- measured: 16 us per
// iteration of a 10-iteration alignment that no kernel and no solver accounted for.
int resolve_sweep_timings(icpgpu_ctx* c, bool block = true) {
  if (c->pending.empty()) return ICPGPU_OK;
  if (block) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    hipEvent_t last = c->ev_ring[(size_t)c->pending.back().slot * 3 + 2];
    hipError_t q = hipErrorNotReady;
    for (int spin = 0; spin < 16 && (q = hipEventQuery(last)) == hipErrorNotReady; ++spin) {
    }
    if (q == hipErrorNotReady) return ICPGPU_OK;
    if (q != hipSuccess) return fail(c, ICPGPU_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
  }
  for (const auto& p : c->pending) {
    float nn_ms = 0.f, red_ms = 0.f;
    hipEvent_t* e = &c->ev_ring[(size_t)p.slot * 3];
    HIP_TRY(c, hipEventElapsedTime(&nn_ms, e[0], e[1]));
    HIP_TRY(c, hipEventElapsedTime(&red_ms, e[1], e[2]));
    if (p.grid) {
      c->prof.grid_ms += nn_ms;
      c->prof.grid_timed += 1;
    } else {
      c->prof.nn_ms += nn_ms;
      c->prof.nn_timed += 1;
    }
    c->prof.reduce_ms += red_ms;
    c->prof.reduce_timed += 1;
    c->dev_ms_accum += (double)nn_ms + (double)red_ms;
  }
  c->pending.clear();
  return ICPGPU_OK;
}

// How long a result may take before the wait gives up (ICPGPU_WAIT_TIMEOUT_MS, default 30 s): a kernel that hangs without
// faulting would otherwise keep the caller spinning for ever.  The context is unusable after a timeout (its stream still
// holds the hung kernel); the caller gets ICPGPU_ERR_HIP instead of a dead thread.
static double wait_timeout_ms() {
  static const double v = [] { const char* e = std::getenv("ICPGPU_WAIT_TIMEOUT_MS"); const double x = e ? std::atof(e) : 0.0; return x > 0.0 ? x : 30000.0; }();
  return v;
}

// the sums mailbox: pair k = {sum bits, sequence number} at words 2k, 2k + 1 (reduce_final_kernel)
bool flags_ready(const volatile unsigned long long* pairs, int n_pairs, unsigned long long seq) {
  bool all = true;
  for (int k = 0; k < n_pairs; ++k) all = all && (pairs[2 * k + 1] == seq);
  return all;
}
// ... into c->h_sums, where everybody reads them
void take_sums(icpgpu_ctx* c) {
  for (int k = 0; k < kReduceTerms; ++k) {
    const unsigned long long bits = c->h_flags[2 * k];
    std::memcpy(&c->h_sums[k], &bits, sizeof bits);
  }
}

// Spin on the mailbox flags until every term of sweep `seq` has landed.  The stream is queried now and then so that a
// faulted kernel turns into an error instead of an endless wait, and the clock so that a hung one does.
int wait_flags(icpgpu_ctx* c, const volatile unsigned long long* flags, int n_flags, unsigned long long seq) {
  std::chrono::steady_clock::time_point t0;
  for (unsigned spins = 1;; ++spins) {
    if (flags_ready(flags, n_flags, seq)) break;
    if ((spins & 0x3FFu) == 0) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {  // everything retired: the flags must be there on the next look
        if (flags_ready(flags, n_flags, seq)) break;
        return fail(c, ICPGPU_ERR_HIP, "reduction finished without publishing its result");
      }
      if (q != hipErrorNotReady) return fail(c, ICPGPU_ERR_HIP, "HIP error while waiting for a reduction: %s", hipGetErrorString(q));
      const auto now = std::chrono::steady_clock::now();
      if (spins == 0x400u) t0 = now;
      else if (std::chrono::duration<double, std::milli>(now - t0).count() > wait_timeout_ms())
        return fail(c, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for a kernel's result (hung kernel?)", wait_timeout_ms());
    }
    if (spins > 8192u) std::this_thread::yield();  // a long (brute-force) sweep: stop monopolising the core
#if defined(__x86_64__)
    else __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ICPGPU_OK;
}

int wait_sums(icpgpu_ctx* c, unsigned long long seq) {
  const int rc = wait_flags(c, c->h_flags, kReduceTerms, seq);
  if (!rc) take_sums(c);
  return rc;
}

// One NN sweep + reduction with transform T, in three steps so that one host thread can keep several contexts busy
// (icpgpu_align_batch): sweep_issue queues the kernels, the 17 sums land in c->h_sums when flags_ready(c->h_flags, ...,
// ticket.seq) -- the host polls the mailbox, it never waits for the stream -- and sweep_complete finishes the rare ungated
// sweep whose grid stage left too many points for the few-queries kernel.
struct SweepTicket {
  unsigned long long seq = 0;
  volatile int* few_host = nullptr;  // ungated grid search: number of points its grid stage left unmatched
  const float4* red_src = nullptr;   // keys path: the array the keys index
  int red_n = 0;
  Xform T{};
  float thr = 0.f;
};

int sweep_issue(icpgpu_ctx* c, const Xform& T, float thr, bool open_range, SweepTicket& tk) {
  static const bool timing = [] { const char* e = std::getenv("ICPGPU_P2P_TIMING"); return e && std::atoi(e) != 0; }();
  if (timing) {
    c->pt_issue_in = std::chrono::steady_clock::now();
    if (c->pt_n) c->pt_solve += std::chrono::duration<double, std::micro>(c->pt_issue_in - c->pt_ready).count();
  }
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  int rc = ensure(c, c->keys, (size_t)(n_s ? n_s : 1) * sizeof(unsigned long long));
  if (rc) return rc;
  auto* keys = static_cast<unsigned long long*>(c->keys.ptr);
  auto* partials = static_cast<double*>(c->partials.ptr);
  double* d_sums = c->h_sums_dev;
  const bool use_grid = grid_ready(c) && n_s > 0 && (open_range || thr <= c->grid.cutoff * c->grid.cutoff);
  if ((int)c->pending.size() >= kEventRing && (rc = resolve_sweep_timings(c))) return rc;
  const int slot = (int)c->pending.size();
  hipEvent_t* ev = &c->ev_ring[(size_t)slot * 3];
  const unsigned long long seq = ++c->sums_seq;
  // kernel timing is sampled: an event record is a barrier packet on the queue, three of them per sweep cost 6-7 us
  const bool timed = c->timing_every <= 1 || (c->sweep_counter++ % (unsigned)c->timing_every) == 0;
#define EVREC(e) do { if (timed) HIP_TRY(c, hipEventRecord((e), c->stream)); } while (0)
  EVREC(ev[0]);
  const float4* red_src = nullptr;
  int red_n = 0;
  volatile int* few_host = nullptr;
  // EXPERIMENTAL (off by default, read per sweep so that a test can switch it): the grid search on the matrix cores,
  // icp_tile.hip -- bit-identical results, faster at 50k x 50k, slower at 200k x 200k (experiments)
  const char* tile_env = std::getenv("ICPGPU_TILE_SEARCH");
  const int tile_search = tile_env ? std::atoi(tile_env) : 0;
  if (use_grid && !open_range && tile_search && source_ordered(c) && n_s >= kMfmaMinPoints) {
    // the grid search on the matrix cores (icp_tile.hip): keys, then the keys-path reduction
    if ((rc = source_in_morton_order(c))) return rc;
    TileSeed& S = c->tile_seed;
    const bool seeded = S.valid && S.src_version == c->src_version && S.grid_version == c->grid.version && S.n_s == n_s;
    if ((rc = ensure(c, S.keys, (size_t)n_s * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, S.prev, (size_t)c->brute_order.n * sizeof(float4)))) return rc;
    unsigned long long* d_stats = nullptr;
    if (tile_search > 1) {
      if ((rc = ensure(c, S.stats, 8 * sizeof(unsigned long long)))) return rc;
      d_stats = static_cast<unsigned long long*>(S.stats.ptr);
      HIP_TRY(c, hipMemsetAsync(d_stats, 0, 8 * sizeof(unsigned long long), c->stream));
    }
    HIP_TRY(c, launch_fill_keys(keys, n_s, c->stream));
    HIP_TRY(c, launch_nn_tile_search(static_cast<const float4*>(c->brute_order.pts.ptr), c->brute_order.n, T,
                                     static_cast<const float4*>(c->grid.sorted.ptr), static_cast<const int*>(c->grid.cell_start.ptr),
                                     c->grid.g, c->tgt.data(), n_t, thr,
                                     seeded ? static_cast<const unsigned long long*>(S.keys.ptr) : nullptr,
                                     static_cast<float4*>(S.prev.ptr), seeded, keys, d_stats, c->stream));
    HIP_TRY(c, hipMemcpyAsync(S.keys.ptr, keys, (size_t)n_s * sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
    S.valid = true;
    S.src_version = c->src_version;
    S.grid_version = c->grid.version;
    S.n_s = n_s;
    if (d_stats) {
      unsigned long long h[8] = {0};
      HIP_TRY(c, hipMemcpyAsync(h, d_stats, sizeof(h), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      const double nb = (double)(h[5] ? h[5] : 1);
      fprintf(stderr, "[icpgpu] tile search (%s): %.3f G pairs, %.0f workgroups, %.0f candidates each; cycles per workgroup: preamble %.0f, "
                      "rows %.0f, tile fills %.0f, steps %.0f; slowest workgroup %.0f\n", seeded ? "seeded" : "cold", (double)h[0] * 1e-9, nb,
              (double)h[6] / nb, (double)h[1] / nb, (double)h[2] / nb, (double)h[3] / nb, (double)h[4] / nb, (double)h[7]);
    }
    EVREC(ev[1]);
    red_src = c->src.data();
    red_n = n_s;
    if ((rc = ensure(c, c->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double)))) return rc;
    partials = static_cast<double*>(c->partials.ptr);
    HIP_TRY(c, launch_reduce(red_src, red_n, c->tgt.data(), keys, T, thr, partials, d_sums, c->h_flags_dev, seq, c->stream));
  } else if (use_grid && !open_range) {
    // cell-ordered source when there is one (non-finite points are absent from it: they never match anyway)
    const bool ordered = source_ordered(c);
    const float4* src_pts = ordered ? static_cast<const float4*>(c->src_grid.sorted.ptr) : c->src.data();
    const int n_q = ordered ? c->src_grid.n_binned : n_s;
    const int blocks = grid_search_blocks(n_q);
    if ((rc = ensure(c, c->partials, (size_t)blocks * kReduceTerms * sizeof(double)))) return rc;
    partials = static_cast<double*>(c->partials.ptr);
    float4* prev = nullptr;
    bool use_prev = false;
    if ((rc = prev_neighbours(c, c->grid, src_pts, n_q, grid_flags(c->grid, ordered), prev, use_prev))) return rc;
    std::chrono::steady_clock::time_point tl0;
    if (timing) tl0 = std::chrono::steady_clock::now();
    HIP_TRY(c, launch_nn_grid_search(src_pts, n_q, grid_flags(c->grid, ordered), T, static_cast<const float4*>(c->grid.sorted.ptr),
                                     static_cast<const int*>(c->grid.cell_start.ptr), c->grid.g, thr, nullptr, partials, nullptr,
                                     nullptr, c->stream, prev, use_prev));
    if (timing) {
      const auto tl1 = std::chrono::steady_clock::now();
      c->pt_prelaunch += std::chrono::duration<double, std::micro>(tl0 - c->pt_issue_in).count();
      c->pt_launch += std::chrono::duration<double, std::micro>(tl1 - tl0).count();
      c->pt_issue_in = tl1;
    }
    EVREC(ev[1]);
    HIP_TRY(c, launch_reduce_final(partials, blocks, /*term_major=*/true, d_sums, c->h_flags_dev, seq, c->stream));
  } else {
    red_src = c->src.data();
    red_n = n_s;
    if (use_grid) {
      // the cell-ordered copy of the source when there is one, as in the gated sweep: the sums do not depend on the order,
      // and the neighbours the last gated sweep left behind (same array) bound this search too
      if (source_ordered(c)) {
        red_src = static_cast<const float4*>(c->src_grid.sorted.ptr);
        red_n = c->src_grid.n_binned;
      }
      few_host = reinterpret_cast<volatile int*>(c->h_sums + 20);  // a spare slot of the mailbox
      *few_host = -1;
      if ((rc = nn_keys_grid(c, c->grid, red_src, red_n, c->tgt.data(), n_t, T, keys, reinterpret_cast<int*>(c->h_sums_dev + 20))))
        return rc;
    } else {
      if ((rc = nn_keys_brute(c, c->tgt.data(), n_t, T, keys))) return rc;
    }
    EVREC(ev[1]);
    if ((rc = ensure(c, c->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double)))) return rc;
    partials = static_cast<double*>(c->partials.ptr);
    HIP_TRY(c, launch_reduce(red_src, red_n, c->tgt.data(), keys, T, thr, partials, d_sums, c->h_flags_dev, seq, c->stream));
  }
  EVREC(ev[2]);
  if (timed) c->pending.push_back({slot, use_grid});
  c->call_sweeps += 1;
  c->call_timed += timed ? 1 : 0;
#undef EVREC
  if (use_grid) {
    c->prof.grid_launches += 1;
    c->prof.grid_bytes += 16ull * ((uint64_t)n_s + (uint64_t)n_t) + (open_range ? 8ull * (uint64_t)n_s : 136ull * (uint64_t)grid_search_blocks(n_s));
  } else {
    c->prof.nn_launches += (n_s > 0);
    c->prof.nn_pairs += (uint64_t)n_s * (uint64_t)n_t;
    c->prof.nn_bytes += 16ull * ((uint64_t)n_s + (uint64_t)n_t) + 8ull * (uint64_t)n_s;
  }
  c->prof.reduce_launches += 1;
  c->prof.reduce_bytes += (use_grid && !open_range) ? 136ull * (uint64_t)grid_search_blocks(n_s) : 40ull * (uint64_t)n_s + 136;
  if (timing) c->pt_rest += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c->pt_issue_in).count();
  tk.seq = seq;
  tk.few_host = few_host;
  tk.red_src = red_src;
  tk.red_n = red_n;
  tk.T = T;
  tk.thr = thr;
  return ICPGPU_OK;
}

bool sweep_ready(const icpgpu_ctx* c, const SweepTicket& tk) { return flags_ready(c->h_flags, kReduceTerms, tk.seq); }

// after the sums of tk have arrived
int sweep_complete(icpgpu_ctx* c, SweepTicket& tk) {
  std::atomic_thread_fence(std::memory_order_acquire);
  take_sums(c);  // (callers that polled sweep_ready themselves have not)
  if (!tk.few_host) return ICPGPU_OK;
  // ungated search: the few-queries kernel completed the unmatched points unless there were too many for it (then the sums
  // just received miss them: tiled brute-force completion and a second reduction)
  const int n_un = *tk.few_host;
  c->prof.grid_fallback_points += (uint64_t)(n_un > 0 ? n_un : 0);
  if (n_un <= kFewQueries) return ICPGPU_OK;
  auto* keys = static_cast<unsigned long long*>(c->keys.ptr);
  int rc = complete_deferred_keys(c, c->grid, tk.red_src, tk.red_n, c->tgt.data(), (int)c->tgt.n, tk.T, keys, n_un);
  if (rc) return rc;
  const unsigned long long seq2 = ++c->sums_seq;
  HIP_TRY(c, launch_reduce(tk.red_src, tk.red_n, c->tgt.data(), keys, tk.T, tk.thr, static_cast<double*>(c->partials.ptr),
                           c->h_sums_dev, c->h_flags_dev, seq2, c->stream));
  c->prof.reduce_launches += 1;
  tk.few_host = nullptr;
  tk.seq = seq2;
  return wait_sums(c, seq2);
}

int nn_and_reduce(icpgpu_ctx* c, const Xform& T, float thr, bool open_range) {
  SweepTicket tk;
  int rc = sweep_issue(c, T, thr, open_range, tk);
  if (rc) return rc;
  if ((rc = wait_sums(c, tk.seq))) return rc;
  return sweep_complete(c, tk);
}


int write_output_cloud(icpgpu_ctx* c, const Xform& T, float* out_xyzw) {
  const int n_s = (int)c->src.n;
  if (!out_xyzw || n_s == 0) return ICPGPU_OK;
  int rc = ensure(c, c->out, (size_t)n_s * sizeof(float4));
  if (rc) return rc;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, launch_transform(c->src.data(), n_s, T, static_cast<float4*>(c->out.ptr), c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(c, hipMemcpyAsync(out_xyzw, c->out.ptr, (size_t)n_s * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  c->prof.transform_launches += 1;
  c->prof.transform_ms += ms;
  c->prof.transform_bytes += 32ull * (uint64_t)n_s;
  return ICPGPU_OK;
}

// pcl::VoxelGrid<PointXYZ>::filter on a device-resident cloud (icp_odometer.cpp:96-101). out receives *n_out points
// (ascending cell index). *passthrough = PCL's "leaf size too small for the input dataset" case: input returned as is.
int voxel_filter_device(icpgpu_ctx* c, const float4* d_in, int n, float leaf, DeviceBuf& out, int* n_out, bool* passthrough) {
  *n_out = 0;
  *passthrough = false;
  if (!(leaf > 0.f) || !std::isfinite(leaf)) return fail(c, ICPGPU_ERR_INVALID_ARG, "voxel filter: leaf size must be positive");
  if (n <= 0) return ICPGPU_OK;
  int rc = ensure(c, c->vox_ints, 16 * sizeof(int));
  if (rc) return rc;
  int* d_ints = static_cast<int*>(c->vox_ints.ptr);
  HIP_TRY(c, launch_bbox(d_in, n, d_ints, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_ints, d_ints, 6 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float lo[3], hi[3];
  decode_bbox(c->h_ints, lo, hi);
  if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) return ICPGPU_OK;  // no finite point
  const float inv = 1.0f / leaf;  // PCL: inverse_leaf_size_ = 1 / leaf_size_ in float
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
  float ms = 0.f;
  // the direct path (one distribution pass + a sort in LDS): every cloud up to 2M points; it reports the rare cloud it cannot
  // take (thousands of points in one voxel) through `status`, and the library-sort path runs instead
  static const bool force_sort = [] { const char* e = std::getenv("ICPGPU_VOXEL_SORT"); return e && std::atoi(e) != 0; }();
  bool done = false;
  // (PCL's overflow test uses the float extents, its cell index the integer ones: when these are one cell wider the index of
  // the topmost cells can wrap in int32 and PCL -- and the sort path, on signed keys -- puts them first.  The direct path's
  // buckets assume keys in [0, number of cells): leave that corner to the sort path.)
  const bool keys_may_wrap = (long long)divb[0] * divb[1] * divb[2] > (long long)INT32_MAX;
  if (!force_sort && !keys_may_wrap && n <= (1 << 21)) {
    if ((rc = ensure(c, c->vox_bins, voxel_direct_scratch_ints(n) * sizeof(int)))) return rc;
    if (c->vox_bins_zeroed != c->vox_bins.ptr || c->vox_bins_zeroed_cap != c->vox_bins.cap) {
      HIP_TRY(c, hipMemsetAsync(c->vox_bins.ptr, 0, c->vox_bins.cap, c->stream));
      c->vox_bins_zeroed = c->vox_bins.ptr;
      c->vox_bins_zeroed_cap = c->vox_bins.cap;
    }
    if ((rc = ensure(c, c->vox_pub, (size_t)voxel_direct_groups(n) * sizeof(unsigned long long)))) return rc;
    if (c->vox_pub_zeroed != c->vox_pub.ptr || c->vox_pub_zeroed_cap != c->vox_pub.cap) {
      HIP_TRY(c, hipMemsetAsync(c->vox_pub.ptr, 0, c->vox_pub.cap, c->stream));
      c->vox_pub_zeroed = c->vox_pub.ptr;
      c->vox_pub_zeroed_cap = c->vox_pub.cap;
    }
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    hipError_t le = launch_voxel_grid_direct(d_in, n, inv, minb, divb, static_cast<int*>(c->vox_bins.ptr),
                                             static_cast<unsigned long long*>(c->vox_pub.ptr),
                                             static_cast<int*>(c->vox_keys.ptr), static_cast<int*>(c->vox_keys.ptr) + n,
                                             static_cast<unsigned long long*>(c->vox_vals.ptr), static_cast<float4*>(out.ptr),
                                             d_ints + 6, d_ints + 8, c->stream);
    if (le != hipSuccess) {
      c->vox_bins_zeroed = nullptr;
      return fail(c, ICPGPU_ERR_HIP, "voxel filter: %s", hipGetErrorString(le));
    }
    HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_ints + 6, d_ints + 6, 3 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    hipError_t se = hipStreamSynchronize(c->stream);
    if (se != hipSuccess) {
      c->vox_bins_zeroed = nullptr;
      return fail(c, ICPGPU_ERR_HIP, "voxel filter: %s", hipGetErrorString(se));
    }
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    done = c->h_ints[8] == 0;
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

void init_result(icpgpu_result* r) {
  std::memset(r, 0, sizeof(*r));
  for (int i = 0; i < 16; ++i) r->T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  r->fitness = NAN;
}

// A point-to-point alignment as a resumable run: begin() queues the first sweep, advance() -- called once the sweep's sums
// have arrived -- does the host's share of an iteration (Umeyama / SVD, convergence test: icp_solver.cpp) and queues the
// next sweep, the fitness sweep, or finishes.  icpgpu_align drives one run to the end; icpgpu_align_batch keeps several
// contexts' runs in flight from one host thread.
struct P2PRun {
  enum Phase { Idle, Iterating, Fitness, Done } phase = Idle;
  Mat4d final_T = mat4_identity();
  ConvergenceCriteria crit{1, 0.0, 0.0, false};
  float thr = 0.f;
  int nr_iter = 0, state = ICPGPU_NOT_CONVERGED, want_fitness = 0;
  bool converged = false;
  unsigned n_corr = 0;
  double mse = 0.0;
  SweepTicket ticket;
  icpgpu_result* res = nullptr;
  float* out_xyzw = nullptr;
  std::chrono::steady_clock::time_point t_start, t_issue;
};

int p2p_finish(icpgpu_ctx* c, P2PRun& r) {
  const Xform Tf = to_xform(r.final_T);
  int rc = write_output_cloud(c, Tf, r.out_xyzw);
  if (rc) return rc;
  if ((rc = resolve_sweep_timings(c, /*block=*/false))) return rc;
  r.res->t_device_ms = c->call_timed ? c->dev_ms_accum * (double)c->call_sweeps / (double)c->call_timed : 0.0;
  r.res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_start).count();
  r.phase = P2PRun::Done;
  return ICPGPU_OK;
}

// everything of p2p_begin that needs no index: the run is Done afterwards when the target is empty, Idle otherwise
int p2p_prepare(icpgpu_ctx* c, P2PRun& r, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  r = P2PRun{};
  r.t_start = std::chrono::steady_clock::now();
  r.res = res;
  r.out_xyzw = out_xyzw;
  r.want_fitness = want_fitness;
  init_result(res);
  c->prof.aligns += 1;
  {
    int rc = resolve_sweep_timings(c, /*block=*/false);
    if (rc) return rc;
    c->dev_ms_accum = 0.0;
    c->call_sweeps = c->call_timed = 0;
    c->prev.valid = c->tile_seed.valid = false;  // every alignment starts cold
  }
  if (guess)
    for (int i = 0; i < 16; ++i) r.final_T[i] = (double)guess[i];

  // pcl::Registration::setInputTarget refuses an empty target, initCompute() then fails and align() returns
  // with converged_ = false and final_transformation_ = identity.
  if (c->tgt.n == 0) {
    r.final_T = mat4_identity();
    c->final_T = r.final_T;
    c->have_final = true;
    return p2p_finish(c, r);
  }

  const icpgpu_params& P = c->params;
  r.crit = ConvergenceCriteria(P.max_iterations, P.transformation_epsilon, P.euclidean_fitness_epsilon, P.force_iterations != 0);
  r.thr = threshold_from(P.max_correspondence_distance * P.max_correspondence_distance);
  return ICPGPU_OK;
}

int p2p_begin(icpgpu_ctx* c, P2PRun& r, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  int rc = p2p_prepare(c, r, guess, out_xyzw, want_fitness, res);
  if (rc || r.phase == P2PRun::Done) return rc;
  if ((rc = ensure_grid(c, r.thr))) return rc;
  if ((rc = ensure_source_order(c, r.thr))) return rc;
  r.phase = P2PRun::Iterating;
  r.t_issue = std::chrono::steady_clock::now();
  return sweep_issue(c, to_xform(r.final_T), r.thr, false, r.ticket);
}

// precondition: sweep_ready(c, r.ticket).  With `deferred` the next GATED sweep is not issued here: *deferred = true tells
// the caller (the lock-step batch path) that the run wants one at r.final_T.
// (*deferred: 0 nothing, 1 a gated sweep, 2 the ungated fitness sweep)
int p2p_advance(icpgpu_ctx* c, P2PRun& r, int* deferred = nullptr) {
  if (deferred) *deferred = 0;
  int rc = sweep_complete(c, r.ticket);
  if (rc) return rc;
  const double* sums = c->h_sums;
  if (r.phase == P2PRun::Fitness) {
    r.res->fitness = sums[0] > 0.0 ? sums[16] / sums[0] : DBL_MAX;
    return p2p_finish(c, r);
  }
  const icpgpu_params& P = c->params;
  bool stop = false;
  r.n_corr = (unsigned)sums[0];
  Mat4d Tk;
  if ((int)r.n_corr < P.min_correspondences || !solve_umeyama(sums, Tk)) {
    r.state = ICPGPU_CONV_NO_CORRESPONDENCES;
    r.converged = false;
    stop = true;
  } else {
    r.final_T = mat4_mul(Tk, r.final_T);
    r.mse = sums[16] / sums[0];
    ++r.nr_iter;
    c->prof.iterations += 1;
    if (r.crit.has_converged(r.nr_iter, Tk, r.mse)) {
      r.converged = true;
      r.state = r.crit.state();
      stop = true;
    }
  }
  r.t_issue = std::chrono::steady_clock::now();
  if (!stop) {
    if (deferred) {
      *deferred = 1;
      return ICPGPU_OK;
    }
    return sweep_issue(c, to_xform(r.final_T), r.thr, false, r.ticket);
  }

  c->final_T = r.final_T;
  c->have_final = true;
  mat4_to_float(r.final_T, r.res->T);
  r.res->converged = r.converged ? 1 : 0;
  r.res->iterations = r.nr_iter;
  r.res->convergence_state = r.state;
  r.res->n_correspondences = r.n_corr;
  r.res->mse_last = r.mse;
  if (r.want_fitness) {
    r.phase = P2PRun::Fitness;
    if (deferred) {
      *deferred = 2;
      return ICPGPU_OK;
    }
    return sweep_issue(c, to_xform(r.final_T), FLT_MAX, true, r.ticket);
  }
  return p2p_finish(c, r);
}

int align_p2p(icpgpu_ctx* c, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  P2PRun r;
  int rc = p2p_begin(c, r, guess, out_xyzw, want_fitness, res);
  static const bool timing = [] { const char* e = std::getenv("ICPGPU_P2P_TIMING"); return e && std::atoi(e) != 0; }();
  while (!rc && r.phase != P2PRun::Done) {
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = wait_sums(c, r.ticket.seq))) break;
    if (timing) {
      c->pt_ready = std::chrono::steady_clock::now();
      c->pt_wait += std::chrono::duration<double, std::micro>(c->pt_ready - t0).count();
      c->pt_n += 1;
    }
    rc = p2p_advance(c, r);
  }
  return rc;
}

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

// per-point covariances of `cloud` (20-NN in its own grid); cached per cloud version
int ensure_covariances(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, GridIndex& G, DeviceBuf& cov, uint64_t& cov_version) {
  if (cov_version == version && cov.ptr) return ICPGPU_OK;
  // the covariance search has no distance cap; the grid only needs cells of a useful size: same rule as the NN grid
  const double cut = std::max(1e-3, c->params.max_correspondence_distance);
  static const double knn_pop = [] { const char* e = std::getenv("ICPGPU_KNN_POP"); return e ? std::atof(e) : 8.0; }();
  int rc = build_grid(c, cloud, version, std::isfinite(cut) ? std::min(cut, 1e6) : 1.0, /*adapt=*/false, G, nullptr, knn_pop);
  if (rc) return rc;
  if (!G.usable) return fail(c, ICPGPU_ERR_UNSUPPORTED, "GICP: cannot index this cloud (degenerate or non-finite input)");
  if ((rc = ensure(c, cov, cloud.n * 6 * sizeof(double)))) return rc;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, launch_gicp_covariances(cloud.data(), (int)cloud.n, static_cast<const float4*>(G.sorted.ptr),
                                     static_cast<const int*>(G.cell_start.ptr), G.g, static_cast<double*>(cov.ptr), c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  c->prof.gicp_cov_launches += 1;
  c->prof.gicp_cov_ms += ms;
  c->prof.gicp_cov_points += (uint64_t)cloud.n;
  cov_version = version;
  return ICPGPU_OK;
}

// (hi, lo) += (bh, bl): the cascaded double-double merge of icp_gicp.hip (TwoSum on the high parts, the small parts in
// plain float64)
static inline void gicp_dd_add(double& hi, double& lo, double bh, double bl) {
  const double s = hi + bh;
  const double bb = s - hi;
  const double e = (hi - (s - bb)) + (bh - bb);
  hi = s;
  lo = (lo + bl) + e;
}

// ---- the resident evaluation server of a BFGS run (gicp_server_kernel) ---------------------------------------------
// A command is four 16-byte chunks {3 floats of T, sequence number}; each chunk is ONE aligned 16-byte store, so the
// device never sees half a chunk, and it acts once all four carry the number it waits for.
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
                                c->gicp_cmd, next, (unsigned int)((c->sums_seq + 1) >> 32), c->stream));
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
// The entries a block publishes (icp_kernels.h: {value, tag} pairs): m, the 13 sums' high parts, sum d2, their low parts.
static const int kGicpEntries[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28};
static bool gicp_tags_ready(const volatile double* mailbox, int n_blocks, unsigned long long seq) {
  const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(mailbox);
  bool all = true;
  for (int b = 0; b < n_blocks; ++b, w += kGicpPartialStride)
    for (int e : kGicpEntries) all = all && (w[2 * e + 1] == seq);
  return all;
}
// 0 = all entries of evaluation `seq` are there, 1 = the stream went idle without them (the server gave up), < 0 = error
static int wait_gicp_tags(icpgpu_ctx* c, int n_blocks, unsigned long long seq, bool server) {
  std::chrono::steady_clock::time_point t0;
  for (unsigned spins = 1;; ++spins) {
    if (gicp_tags_ready(c->h_gicp, n_blocks, seq)) break;
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

int align_gicp(icpgpu_ctx* c, const float* guess_in, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  struct ServerGuard {  // whatever way this function is left, no server stays behind
    icpgpu_ctx* c;
    ~ServerGuard() { gicp_server_stop(c); }
  } server_guard{c};
  const auto t_start = std::chrono::steady_clock::now();
  init_result(res);
  c->prof.aligns += 1;
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  const icpgpu_params& P = c->params;
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
  if ((rc = ensure_covariances(c, c->src, c->src_version, c->cov_grid_src, c->cov_src, c->cov_src_version))) return rc;
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
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    if (grid_ready(c)) {
      float4* prev = nullptr;  // each outer iteration's neighbours bound the next one's search
      bool use_prev = false;
      int prc = prev_neighbours(c, c->grid, c->src.data(), n_s, grid_flags(c->grid, false), prev, use_prev);
      if (prc) return prc;
      HIP_TRY(c, launch_nn_grid_search(c->src.data(), n_s, grid_flags(c->grid, false), Tq, static_cast<const float4*>(c->grid.sorted.ptr),
                                       static_cast<const int*>(c->grid.cell_start.ptr), c->grid.g, thr, keys, nullptr, nullptr,
                                       nullptr, c->stream, prev, use_prev));
    } else {
      if ((rc = nn_keys_brute(c, c->tgt.data(), n_t, Tq, keys))) return rc;
    }
    HIP_TRY(c, launch_gicp_mahalanobis(n_s, keys, thr_excl, R, static_cast<const double*>(c->cov_src.ptr),
                                       static_cast<const double*>(c->cov_tgt.ptr), maha, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));

    // rigid_transformation_estimation_: BFGS over x = (t, roll, pitch, yaw), every evaluation one reduction on the device
    double m_count = 0.0;
    auto eval = [&](const Vec6& x, bool want_gradient, GicpEval& out) -> bool {
      float T[16];
      std::memcpy(T, guess, sizeof(T));
      gicp_apply_state(T, x);
      // ~300 evaluations per align, each a dependent launch: ONE kernel of a few workgroups whose partial sums land in the
      // polled host mailbox; the host adds them in workgroup order (deterministic)
      const auto t_eval0 = std::chrono::steady_clock::now();
      unsigned long long seq = ++c->sums_seq;
      if ((unsigned int)seq == kGicpServerExit) seq = (c->sums_seq += 2);  // (never a command number; the server skips it too)
      const int nblk = gicp_direct_blocks(n_s, c->gicp_blocks_most);
      bool have = false;
      static const bool timing = [] { const char* e = std::getenv("ICPGPU_GICP_TIMING"); return e && std::atoi(e) != 0; }();
      std::chrono::steady_clock::time_point tq0, tq1, tq2;
      if (timing) {
        tq0 = std::chrono::steady_clock::now();
        if (c->gt_n && c->gicp_server_on) c->gt_between += std::chrono::duration<double, std::micro>(tq0 - c->gt_last).count();
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
                                    c->h_gicp_dev, c->h_gicp_flags_dev, seq, c->stream) != hipSuccess)
          return false;
        if (wait_gicp_tags(c, nblk, seq, /*server=*/false) != 0) return false;
      }
      {  // workgroup by workgroup, in double-double like the kernel (icp_gicp.hip): the 13 sums are rounded once, here
        double m = 0.0, d2 = 0.0, hi[13] = {}, lo[13] = {};
        const double* part = c->h_gicp;
        for (int b = 0; b < nblk; ++b, part += kGicpPartialStride) {  // (entry e = doubles [2e] value, [2e + 1] tag)
          m += part[0];
          d2 += part[2 * 14];
          for (int k = 0; k < 13; ++k) gicp_dd_add(hi[k], lo[k], part[2 * (1 + k)], part[2 * (16 + k)]);
        }
        c->h_sums[0] = m;
        c->h_sums[14] = d2;
        for (int k = 0; k < 13; ++k) c->h_sums[1 + k] = hi[k] + lo[k];
      }
      if (timing && have) {
        const auto tq3 = std::chrono::steady_clock::now();
        c->gt_cmd += std::chrono::duration<double, std::micro>(tq1 - tq0).count();
        c->gt_wait += std::chrono::duration<double, std::micro>(tq2 - tq1).count();
        c->gt_merge += std::chrono::duration<double, std::micro>(tq3 - tq2).count();
        c->gt_dev_wait += c->h_gicp[2 * 30];   // block 0's stamps (icp_gicp.hip): polling, then work, in microseconds
        c->gt_dev_work += c->h_gicp[2 * 31];
        c->gt_n += 1;
        c->gt_last = tq3;
      }
      c->prof.gicp_cost_launches += 1;
      c->prof.gicp_eval_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_eval0).count();
      const double* s = c->h_sums;
      c->prof.gicp_eval_corr += (uint64_t)s[0];
      m_count = s[0];
      mse = s[0] > 0 ? s[14] / s[0] : 0.0;
      if (!(s[0] >= 1.0)) {
        out.f = 0.0;
        out.g.fill(0.0);
        return true;
      }
      out.f = s[1] / s[0];
      if (want_gradient) {
        const double sc = 2.0 / s[0];
        double Rm[9];
        for (int k = 0; k < 3; ++k) out.g[k] = s[2 + k] * sc;
        for (int k = 0; k < 9; ++k) Rm[k] = s[5 + k] * sc;
        gicp_rotation_gradient(x, Rm, out.g);
      }
      return true;
    };
    // the ~35 dependent evaluations of this outer iteration go to a resident kernel (queued behind the two kernels above)
    if ((rc = gicp_server_start(c, n_s, keys, thr_excl, base, maha))) return rc;
    // number of correspondences (one evaluation at the current state; it is also the BFGS start, cached by the solver)
    Vec6 x = gicp_state_from_matrix(transformation);
    GicpEval probe;
    if (!eval(x, true, probe)) return fail(c, ICPGPU_ERR_HIP, "GICP cost evaluation failed: %s", hipGetErrorString(hipGetLastError()));
    n_corr = (unsigned)m_count;
    std::memcpy(previous, transformation, sizeof(previous));
    if (n_corr < 4) {  // NotEnoughPointsException -> the loop breaks with converged_ = false
      state = ICPGPU_CONV_NO_CORRESPONDENCES;
      break;
    }
    const GicpSolve sr = gicp_minimize(eval, x, 20, 1e-2, &probe);
    gicp_server_stop(c);
    if (sr == GicpSolve::DeviceError) return fail(c, ICPGPU_ERR_HIP, "GICP cost evaluation failed: %s", hipGetErrorString(hipGetLastError()));
    if (sr != GicpSolve::Ok) {  // SolverDidntConvergeException
      state = ICPGPU_NOT_CONVERGED;
      break;
    }
    mat4f_identity(transformation);
    gicp_apply_state(transformation, x);
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    dev_ms += ms;
    c->prof.grid_launches += grid_ready(c) ? 1 : 0;
    c->prof.grid_ms += grid_ready(c) ? ms : 0.0;
    double delta = 0.0;
    for (int k = 0; k < 4; ++k)
      for (int l = 0; l < 4; ++l) {
        const double ratio = (k < 3 && l < 3) ? 1.0 / rot_eps : 1.0 / P.transformation_epsilon;
        delta = std::max(delta, ratio * std::fabs((double)previous[l * 4 + k] - (double)transformation[l * 4 + k]));
      }
    ++nr;
    c->prof.iterations += 1;
    if (nr >= P.max_iterations || (delta < 1 && !P.force_iterations)) {
      converged = true;
      state = nr >= P.max_iterations ? ICPGPU_CONV_ITERATIONS : ICPGPU_CONV_TRANSFORM;
      std::memcpy(previous, transformation, sizeof(previous));
    }
  }
  // PCL's own composition of the result: R = previous.R * guess.R, t = previous.t + guess.t
  float fin[16];
  mat4f_identity(fin);
  for (int r = 0; r < 3; ++r) {
    for (int cc = 0; cc < 3; ++cc) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += previous[k * 4 + r] * guess[cc * 4 + k];
      fin[cc * 4 + r] = s;
    }
    fin[12 + r] = previous[12 + r] + guess[12 + r];
  }
  std::memcpy(res->T, fin, sizeof(fin));
  for (int i = 0; i < 16; ++i) c->final_T[i] = (double)fin[i];
  c->have_final = true;
  res->converged = converged ? 1 : 0;
  res->iterations = nr;
  res->convergence_state = state;
  res->n_correspondences = n_corr;
  res->mse_last = mse;
  const Xform Tf = xform_from_f16(fin);
  if (want_fitness) {
    if ((rc = resolve_sweep_timings(c))) return rc;
    c->dev_ms_accum = 0.0;
    if ((rc = nn_and_reduce(c, Tf, FLT_MAX, true))) return rc;
    if ((rc = resolve_sweep_timings(c))) return rc;
    dev_ms += c->dev_ms_accum;  // 0 when this sweep was not a timed one
    res->fitness = c->h_sums[0] > 0.0 ? c->h_sums[16] / c->h_sums[0] : DBL_MAX;
  }
  if ((rc = write_output_cloud(c, Tf, out_xyzw))) return rc;
  res->t_device_ms = dev_ms;
  res->t_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  return ICPGPU_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" {

int icpgpu_version(void) { return ICPGPU_VERSION_MAJOR * 1000 + ICPGPU_VERSION_MINOR; }

void icpgpu_default_params(icpgpu_params* p) {
  if (!p) return;
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
}

int icpgpu_create(icpgpu_ctx** out_ctx, int device_id) {
  if (!out_ctx) return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "out_ctx is null");
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, ICPGPU_ERR_NO_DEVICE, "no HIP device available (%s); libicpgpu has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device_id < 0 || device_id >= count)
    return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "device_id %d out of range [0, %d)", device_id, count);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  if (e != hipSuccess) return fail(nullptr, ICPGPU_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, ICPGPU_ERR_NO_DEVICE, "device %d is %s; libicpgpu is built for gfx950 (MI355X) only", device_id,
                prop.gcnArchName);

  icpgpu_ctx* c = new (std::nothrow) icpgpu_ctx();
  if (!c) return fail(nullptr, ICPGPU_ERR_OOM, "out of host memory");
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  icpgpu_default_params(&c->params);
  if (const char* v = std::getenv("ICPGPU_NN_VARIANT")) c->nn_variant = std::atoi(v);

  auto bail = [&](const char* what, hipError_t err) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(err);
    icpgpu_destroy(c);
    return fail(nullptr, err == hipErrorOutOfMemory ? ICPGPU_ERR_OOM : ICPGPU_ERR_HIP, "%s", msg.c_str());
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
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
    const size_t n_flags_end = (size_t)kGicpDirectBlocks * kGicpPartialStride + 8 + kGicpDirectBlocks;  // partials, gap, flags
    const size_t cmd_off = (n_flags_end + 7) & ~(size_t)7;                                        // 64-byte aligned
    const size_t n_d = cmd_off + 8;                                                              // + the server's command line
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&c->h_gicp), n_d * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) !=
        hipSuccess)
      return bail("hipHostMalloc", e);
    std::memset(c->h_gicp, 0, n_d * sizeof(double));
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_gicp_dev), c->h_gicp, 0)) != hipSuccess)
      return bail("hipHostGetDevicePointer", e);
    const size_t off = (size_t)kGicpDirectBlocks * kGicpPartialStride + 8;
    c->h_gicp_flags = reinterpret_cast<volatile unsigned long long*>(c->h_gicp + off);
    c->h_gicp_flags_dev = reinterpret_cast<unsigned long long*>(c->h_gicp_dev + off);
    (void)cmd_off;
  }
  if (gicp_server_enabled()) {  // no such memory (no large BAR): the evaluations stay single launches
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&c->gicp_cmd), 4096, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      c->gicp_cmd = nullptr;
    } else if ((e = hipMemset(c->gicp_cmd, 0, 4096)) != hipSuccess) {
      return bail("hipMemset", e);
    }
  }
  c->ev_ring.assign((size_t)kEventRing * 3, nullptr);
  for (auto& ev : c->ev_ring)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
  c->pending.reserve(kEventRing);
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

int icpgpu_destroy(icpgpu_ctx* c) {
  if (c && c->pt_n)
    fprintf(stderr, "[icpgpu] point-to-point sweeps waited for: %llu; per sweep: waiting for the sums %.2f us, sums -> sweep_issue (take, solve, "
                    "convergence) %.2f us, sweep_issue up to the search launch %.2f us, the launch call %.2f us, the rest of sweep_issue "
                    "(final-reduction launch, events) %.2f us\n",
            c->pt_n, c->pt_wait / c->pt_n, c->pt_solve / c->pt_n, c->pt_prelaunch / c->pt_n, c->pt_launch / c->pt_n, c->pt_rest / c->pt_n);
  if (c && c->gt_n)
    fprintf(stderr, "[icpgpu] GICP evaluations through the server: %llu; per evaluation: command write %.2f us, wait for the flags %.2f us "
                    "(device: polling %.2f us, work %.2f us), merge %.2f us, solver between evaluations %.2f us\n",
            c->gt_n, c->gt_cmd / c->gt_n, c->gt_wait / c->gt_n, c->gt_dev_wait / c->gt_n, c->gt_dev_work / c->gt_n, c->gt_merge / c->gt_n,
            c->gt_between / c->gt_n);
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
  if (c->h_ints) (void)hipHostFree(c->h_ints);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_ring)
    if (ev) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ICPGPU_OK;
}

const char* icpgpu_last_error(const icpgpu_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int icpgpu_set_params(icpgpu_ctx* c, const icpgpu_params* p) {
  if (!c || !p) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (p->method != ICPGPU_P2P_SVD && p->method != ICPGPU_GICP) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad method");
  if (p->nn_mode < ICPGPU_NN_AUTO || p->nn_mode > ICPGPU_NN_GRID) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad nn_mode");
  if (p->brute_variant < 0 || p->brute_variant > 2) return fail(c, ICPGPU_ERR_INVALID_ARG, "bad brute_variant");
  c->params = *p;
  return ICPGPU_OK;
}

int icpgpu_get_params(const icpgpu_ctx* c, icpgpu_params* p) {
  if (!c || !p) return ICPGPU_ERR_INVALID_ARG;
  *p = c->params;
  return ICPGPU_OK;
}

#define ENTER(c)                                                   \
  if (!(c)) return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "null context"); \
  HIP_TRY((c), hipSetDevice((c)->device))

int icpgpu_set_source(icpgpu_ctx* c, const float* xyzw, size_t n) {
  ENTER(c);
  c->src_version++;
  return set_cloud_host(c, c->src, xyzw, n);
}
static int promote_internal(icpgpu_ctx* c);

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

unsigned long long icpgpu_fingerprint(const float* xyzw, size_t n) {
  unsigned long long s0 = 0, s1 = 0;
  const unsigned char* b = reinterpret_cast<const unsigned char*>(xyzw);
  for (size_t i = 0; i < n; ++i) {  // (two independent multiply chains per point: ~5 GB/s on one core)
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

static int promote_internal(icpgpu_ctx* c) {
  if (!c->src.set) return fail(c, ICPGPU_ERR_NO_INPUT, "promote_source_to_target: no source set");
  std::swap(c->src, c->tgt);
  c->tgt_fp = c->src_fp;  // (versions are re-stamped below)
  const bool fp_follows = c->src_fp_version == c->src_version;
  // the source's GICP covariances stay valid for the cloud that is now the target
  std::swap(c->cov_src, c->cov_tgt);
  std::swap(c->cov_grid_src, c->cov_grid_tgt);
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
  c->cov_src_version = 0;
  c->src_version++;
  c->src.n = 0;
  c->src.set = false;
  if (c->src.buf.external) c->src.buf = DeviceBuf{};
  c->have_final = false;
  return ICPGPU_OK;
}

int icpgpu_align(icpgpu_ctx* c, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  ENTER(c);
  if (!res) return fail(c, ICPGPU_ERR_INVALID_ARG, "result is null");
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "align: source and target must be set first");
  if (c->params.method == ICPGPU_GICP) return align_gicp(c, guess, out_xyzw, want_fitness, res);
  return align_p2p(c, guess, out_xyzw, want_fitness, res);
}

int icpgpu_fitness(icpgpu_ctx* c, double max_range, double* out) {
  ENTER(c);
  if (!out) return fail(c, ICPGPU_ERR_INVALID_ARG, "out is null");
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "fitness: source and target must be set first");
  const Mat4d T = c->have_final ? c->final_T : mat4_identity();
  int rc = ensure_grid(c, threshold_from(c->params.max_correspondence_distance * c->params.max_correspondence_distance));
  if (rc) return rc;
  rc = nn_and_reduce(c, to_xform(T), threshold_from(max_range), true);
  if (rc) return rc;
  *out = c->h_sums[0] > 0.0 ? c->h_sums[16] / c->h_sums[0] : DBL_MAX;
  return ICPGPU_OK;
}

// ---- icpgpu_align_batch: independent scan pairs ----------------------------------------------------------------------------
// CPUs this process may use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes grant 16 of 256 CPUs).
static int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
#if defined(__linux__)
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char q[64];
    long long period = 0;
    if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
      n = std::min(n, std::max(1, (int)(std::atoll(q) / period)));
    std::fclose(f);
  } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    long long quota = -1, period = 0;
    if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
    std::fclose(g);
    if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(h, "%lld", &period) != 1) period = 0;
      std::fclose(h);
    }
    if (quota > 0 && period > 0) n = std::min(n, std::max(1, (int)(quota / period)));
  }
#endif
  return n;
}

// Host threads of a batch: every one of them spins (on mailboxes, or inside a BFGS run), so there must be no more of them
// than CPUs -- across ALL the processes of the job: one process per GPU is the deployment (LOCAL_WORLD_SIZE, set by
// torch.distributed.run, says how many share this host's CPUs).  ICPGPU_BATCH_THREADS overrides.
// batch drivers inside THIS process that share its CPUs with this context's batch (icpgpu_align_batch_multi: one per device;
// a property of the context, so that concurrent callers cannot overwrite each other's share)
extern "C++" {
namespace icpgpu {
void set_ctx_host_share(icpgpu_ctx* c, int peers) {
  if (c) c->host_share = peers < 1 ? 1 : peers;
}
}
}
static size_t batch_threads(const icpgpu_ctx* c, size_t cap) {
  size_t t = 0;
  if (const char* v = std::getenv("ICPGPU_BATCH_THREADS")) t = (size_t)std::max(0, std::atoi(v));
  else if (const char* w = std::getenv("ICPGPU_BATCH_WORKERS")) t = (size_t)std::max(0, std::atoi(w));  // round-1 name
  if (t == 0) {
    int local = 1;
    if (const char* l = std::getenv("LOCAL_WORLD_SIZE")) local = std::max(1, std::atoi(l));
    local *= c->host_share;
    t = (size_t)std::max(1, usable_cpus() / local);
    t = std::min(t, cap);
  }
  return std::max<size_t>(1, t);
}

// Point-to-point ICP: T host threads, each driving K worker contexts (own stream, scratch, grid, mailbox) ROUND-ROBIN --
// it polls the mailboxes of its contexts and does the host's share of an iteration (3x3 SVD, convergence test, next launch)
// for whichever has answered, so the kernels of K alignments are in flight per thread and nobody blocks on one result.
// T x K = 8 alignments in flight (a 50k-point sweep fills less than half of the chip), T from the CPUs this process may
// use: 4 x 2 on an unshared 16-CPU box, 2 x 4 when eight ranks share it.  GICP: one alignment per thread (its BFGS loop is
// a blocking host loop), T threads.  Every pair is solved exactly as icpgpu_align would solve it.
int icpgpu_align_batch(icpgpu_ctx* c, size_t n_pairs, const float* const* src, const size_t* n_src,
                       const float* const* tgt, const size_t* n_tgt, int want_fitness, icpgpu_result* results) {
  ENTER(c);
  if (n_pairs && (!src || !n_src || !tgt || !n_tgt || !results)) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (n_pairs == 0) return ICPGPU_OK;
  const bool gicp = c->params.method == ICPGPU_GICP;
  // Point-to-point batches run in LOCK-STEP (ICPGPU_BATCH_LOCKSTEP=0: the round-robin scheduler of round 2): a host thread
  // leads a group of `depth` pairs on ONE stream -- their index builds go through their host round trips together, and
  // every ICP iteration of the whole group is one search launch (pair = blockIdx.y, nn_quad_batch_kernel) + one final
  // reduction launch (17 x K workgroups) + K host solves, instead of K x (launch + reduce + poll): ~35 launches per K
  // pairs where there were ~35 per pair.  Two threads keep the GPU fed (one copies its next group in while the other's
  // group iterates); results are bit-identical to icpgpu_align's (same kernels' bodies, same workgroup -> point mapping).
  static const bool lockstep_on = [] { const char* e = std::getenv("ICPGPU_BATCH_LOCKSTEP"); return !e || std::atoi(e) != 0; }();
  const bool lockstep = lockstep_on && !gicp;
  size_t n_threads = batch_threads(c, gicp ? 8 : 4), depth = 1;
  if (!gicp) {
    if (const char* v = std::getenv("ICPGPU_BATCH_DEPTH")) depth = (size_t)std::max(1, std::atoi(v));
    else depth = lockstep ? 8 : std::max<size_t>(2, (8 + n_threads - 1) / n_threads);
    if (lockstep) depth = std::min<size_t>(depth, (size_t)kBatchMax);
  }
  n_threads = std::min(n_threads, n_pairs);
  depth = std::min(depth, (n_pairs + n_threads - 1) / n_threads);
  const size_t n_ctx = n_threads * depth;
  while (c->workers.size() < n_ctx) {
    icpgpu_ctx* w = nullptr;
    const int rc = icpgpu_create(&w, c->device);
    if (rc != ICPGPU_OK) return fail(c, rc, "align_batch: worker context: %s", icpgpu_last_error(nullptr));
    c->workers.push_back(w);
  }
  std::atomic<size_t> next{0};
  std::atomic<bool> abort{false};
  struct ThreadError {  // one slot per host thread: nothing shared is written while the threads run
    int code = ICPGPU_OK;
    size_t pair = 0;
    std::string msg;
  };
  std::vector<ThreadError> errors(n_threads);
  auto load_pair = [&](icpgpu_ctx* w, size_t k, bool sync = true) {  // (the caller's buffers outlive this call: sync is optional)
    w->src_version++;
    int rc = set_cloud_host(w, w->src, src[k], n_src[k], sync);
    w->tgt_version++;
    if (!rc) rc = set_cloud_host(w, w->tgt, tgt[k], n_tgt[k], sync);
    return rc;
  };
  auto work = [&](size_t t) {
    ThreadError& err = errors[t];
    auto failed = [&](int rc, size_t k, icpgpu_ctx* w) {
      err.code = rc;
      err.pair = k;
      err.msg = w->err;
      abort.store(true);
    };
    if (hipSetDevice(c->device) != hipSuccess) {
      err.code = ICPGPU_ERR_HIP;
      err.msg = "hipSetDevice failed in a batch thread";
      abort.store(true);
      return;
    }
    icpgpu_ctx* const* ws = &c->workers[t * depth];
    for (size_t s = 0; s < depth; ++s) {
      ws[s]->params = c->params;
      ws[s]->nn_variant = c->nn_variant;
      // Every GICP worker's BFGS runs keep its workgroups resident and a host thread spinning.  8 workers fit the chip
      // with room for everybody's searches; 16 were measured 5x SLOWER than single launches (servers wait for slots other
      // servers hold until their 50 ms patience runs out).
      ws[s]->gicp_server_allowed = n_threads <= kMaxServerWorkers;
      // ... and each worker's evaluations get their share of the ~512 workgroups of that size the chip holds at once (64
      // apiece for 8 workers, as measured in round 1; a lone alignment uses up to 256)
      ws[s]->gicp_blocks_most = std::max(16, std::min(kGicpDirectBlocks, 512 / (int)std::max<size_t>(1, n_threads)));
    }
    if (gicp) {  // one blocking alignment after the other
      icpgpu_ctx* w = ws[0];
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= n_pairs || abort.load()) return;
        int rc = load_pair(w, k);
        if (!rc) rc = align_gicp(w, nullptr, nullptr, want_fitness, &results[k]);
        if (rc) return failed(rc, k, w);
      }
    }
    const double timeout_ms = wait_timeout_ms();
    if (lockstep) {
      // ---- lock-step groups ----------------------------------------------------------------------------------------
      struct Slot {
        icpgpu_ctx* w = nullptr;
        size_t pair = 0;
        P2PRun run;
        GridBuild gb;
        bool lock = false;     // iterates inside the group's lock-step launches (else: the single-pair state machine)
        bool wants = false;    // lock-step: its next gated sweep is due
        bool wants_fit = false;  // lock-step: its ungated fitness sweep is due
        bool waiting = false;  // lock-step: a sweep of it is in flight
        bool pack = false;
        float4* prev = nullptr;
      };
      std::vector<Slot> slots(depth);
      icpgpu_ctx* lead = ws[0];
      hipStream_t gstream = lead->stream;
      std::vector<hipStream_t> own(depth);
      for (size_t s2 = 0; s2 < depth; ++s2) {
        slots[s2].w = ws[s2];
        own[s2] = ws[s2]->stream;
        ws[s2]->stream = gstream;  // one queue for the group: builds, sweeps and fitness sweeps are ordered by it
      }
      struct Restore {
        std::vector<hipStream_t>& own;
        icpgpu_ctx* const* ws;
        hipStream_t g;
        ~Restore() {
          (void)hipStreamSynchronize(g);
          for (size_t i = 0; i < own.size(); ++i) ws[i]->stream = own[i];
        }
      } restore{own, ws, gstream};
      if (ensure(lead, lead->batch_table, depth * sizeof(BatchPair))) return failed(ICPGPU_ERR_OOM, 0, lead);
      std::vector<BatchPair> table(depth);
      auto sync_or_fail = [&](Slot& sl) {
        const hipError_t e = hipStreamSynchronize(gstream);
        if (e == hipSuccess) return true;
        fail(sl.w, ICPGPU_ERR_HIP, "lock-step batch: %s", hipGetErrorString(e));
        failed(ICPGPU_ERR_HIP, sl.pair, sl.w);
        return false;
      };
      int timed_pairs = 0;          // pairs of the launch whose events are outstanding
      unsigned step_counter = 0;
      static const bool bt_on = [] { const char* e = std::getenv("ICPGPU_BATCH_TIMING"); return e && std::atoi(e) != 0; }();
      double bt_fill = 0, bt_build = 0, bt_iter = 0;
      size_t bt_groups = 0, bt_steps = 0;
      struct BtPrint {
        const bool& on; double &f, &b, &i; size_t &g, &st; size_t t;
        ~BtPrint() { if (on && g) fprintf(stderr, "[icpgpu] batch thread %zu: %zu groups, %zu steps; per group: fill (H2D) %.3f ms, index builds %.3f ms, iterations + fitness %.3f ms\n", t, g, st, f / g, b / g, i / g); }
      } bt_print{bt_on, bt_fill, bt_build, bt_iter, bt_groups, bt_steps, t};
      for (;;) {
        const auto bt0 = std::chrono::steady_clock::now();
        // (1) fill the group
        size_t n_slots = 0;
        while (n_slots < depth && !abort.load()) {
          const size_t k = next.fetch_add(1);
          if (k >= n_pairs) break;
          Slot& sl = slots[n_slots];
          sl.pair = k;
          sl.lock = sl.wants = sl.wants_fit = sl.waiting = false;
          int rc = load_pair(sl.w, k, /*sync=*/false);
          if (!rc) rc = p2p_prepare(sl.w, sl.run, nullptr, nullptr, want_fitness, &results[k]);
          if (rc) return failed(rc, k, sl.w);
          ++n_slots;
        }
        if (n_slots == 0 || abort.load()) return;
        const auto bt1 = std::chrono::steady_clock::now();
        // (2) the target grids, every build's host round trips shared by the group
        for (size_t i = 0; i < n_slots; ++i) {
          Slot& sl = slots[i];
          sl.gb = GridBuild{};
          if (sl.run.phase == P2PRun::Done) continue;  // empty target
          icpgpu_ctx* w = sl.w;
          const int mode = w->params.nn_mode;
          const bool want = mode == ICPGPU_NN_GRID || (mode == ICPGPU_NN_AUTO && w->tgt.n >= kGridMinTarget);
          const double cut = std::sqrt((double)sl.run.thr) * (1.0 + 1e-6);
          if (!want || !(sl.run.thr > 0.f) || !std::isfinite(cut) || cut > 1e6) {
            w->grid.usable = w->grid.built = false;
            continue;
          }
          const int rc = gb_begin(w, sl.gb, w->tgt, w->tgt_version, cut, /*adapt=*/true, w->grid);
          if (rc) return failed(rc, sl.pair, w);
        }
        for (;;) {
          bool pending = false;
          for (size_t i = 0; i < n_slots; ++i) pending = pending || slots[i].gb.state != GridBuild::Done;
          if (!pending) break;
          if (!sync_or_fail(slots[0])) return;
          for (size_t i = 0; i < n_slots; ++i)
            if (slots[i].gb.state != GridBuild::Done) {
              const int rc = gb_advance(slots[i].w, slots[i].gb);
              if (rc) return failed(rc, slots[i].pair, slots[i].w);
            }
        }
        const auto bt2 = std::chrono::steady_clock::now();
        // (3) who can iterate in lock-step; the others start their own first sweep
        size_t live = 0;
        for (size_t i = 0; i < n_slots; ++i) {
          Slot& sl = slots[i];
          if (sl.run.phase == P2PRun::Done) continue;
          icpgpu_ctx* w = sl.w;
          ++live;
          const int n_s = (int)w->src.n;
          const int flags = grid_flags(w->grid, false);
          sl.lock = grid_ready(w) && n_s > 0 && w->src.n < kOrderSourceMin && source_order_mode() != 1 &&
                    grid_search_batchable(n_s, flags) && sl.run.thr <= w->grid.cutoff * w->grid.cutoff;
          sl.run.phase = P2PRun::Iterating;
          sl.run.t_issue = std::chrono::steady_clock::now();
          if (!sl.lock) {
            int rc = ensure_source_order(w, sl.run.thr);
            if (!rc) rc = sweep_issue(w, to_xform(sl.run.final_T), sl.run.thr, false, sl.run.ticket);
            if (rc) return failed(rc, sl.pair, w);
            continue;
          }
          if (w->src_grid.version != w->src_version) w->src_grid.built = w->src_grid.usable = false;
          const int blocks = grid_search_blocks(n_s);
          int rc = ensure(w, w->partials, (size_t)blocks * kReduceTerms * sizeof(double));
          bool use_prev = false;
          if (!rc) rc = prev_neighbours(w, w->grid, w->src.data(), n_s, flags, sl.prev, use_prev);  // (allocates; the first sweep is cold)
          if (rc) return failed(rc, sl.pair, w);
          w->prev.valid = w->tile_seed.valid = false;
          sl.pack = (flags & kGridPackShortRows) != 0;
          if (!rc) rc = ensure(w, w->keys, (size_t)n_s * sizeof(unsigned long long));
          if (!rc) rc = ensure(w, w->grid.unmatched, (size_t)(n_s + 1) * sizeof(int));
          if (rc) return failed(rc, sl.pair, w);
          BatchPair& bp = table[i];
          bp.keys = static_cast<unsigned long long*>(w->keys.ptr);
          bp.unmatched = static_cast<int*>(w->grid.unmatched.ptr);
          bp.unmatched_count = bp.unmatched + n_s;
          bp.r_max_open = std::min(4 * w->grid.g.r_max, 48);
          bp.src = w->src.data();
          bp.sorted = static_cast<const float4*>(w->grid.sorted.ptr);
          bp.cell_start = static_cast<const int*>(w->grid.cell_start.ptr);
          bp.partials = static_cast<double*>(w->partials.ptr);
          bp.prev_nn = sl.prev;
          bp.flags = w->h_flags_dev;
          bp.g = w->grid.g;
          bp.accept_thr = sl.run.thr;
          bp.n_s = n_s;
          bp.qpw = grid_search_qpw(n_s);
          bp.xcd_map = 0;
          bp.blocks = blocks;
          sl.wants = true;
        }
        // one row-walk variant for the whole group (the packed walk of sparse targets is a speed choice, the neighbours are the
        // same): the majority's, so that a step is ONE launch
        {
          int n_lock = 0, n_pack = 0;
          for (size_t i = 0; i < n_slots; ++i)
            if (slots[i].lock) {
              ++n_lock;
              n_pack += slots[i].pack ? 1 : 0;
            }
          const bool group_pack = 2 * n_pack >= n_lock && n_pack > 0;
          for (size_t i = 0; i < n_slots; ++i) slots[i].pack = group_pack;
        }
        if (hipMemcpyAsync(lead->batch_table.ptr, table.data(), n_slots * sizeof(BatchPair), hipMemcpyHostToDevice, gstream) != hipSuccess) {
          fail(lead, ICPGPU_ERR_HIP, "lock-step batch: table upload");
          return failed(ICPGPU_ERR_HIP, slots[0].pair, lead);
        }
        // (4) iterate: one search launch + one reduction launch per step and row-walk class for the whole group
        unsigned idle_spins = 0;
        while (live > 0) {
          // a step is launched when no lock-step sweep of the group is in flight any more (the step IS the batch; pairs in
          // their fitness sweep or on the single-pair path do not hold it up)
          bool lock_in_flight = false;
          for (size_t i = 0; i < n_slots; ++i) lock_in_flight = lock_in_flight || (slots[i].lock && slots[i].waiting);
          for (int kind = 0; kind < 2 && !lock_in_flight; ++kind) {  // 0: the gated sweeps due, 1: the fitness sweeps due
            BatchStep step;
            int n_act = 0, max_blocks = 0;
            bool group_pack = false;
            step.use_prev_mask = 0u;
            for (size_t i = 0; i < n_slots; ++i) {
              Slot& sl = slots[i];
              if (!sl.lock || !(kind == 0 ? sl.wants : sl.wants_fit)) continue;
              icpgpu_ctx* w = sl.w;
              group_pack = sl.pack;
              bool use_prev = false;
              float4* buf = nullptr;
              if (prev_neighbours(w, w->grid, w->src.data(), (int)w->src.n, grid_flags(w->grid, false), buf, use_prev) || buf != sl.prev) {
                fail(w, ICPGPU_ERR_HIP, "lock-step batch: previous-neighbour buffer moved");
                return failed(ICPGPU_ERR_HIP, sl.pair, w);
              }
              step.T[n_act] = to_xform(sl.run.final_T);
              step.seq[n_act] = ++w->sums_seq;
              step.slot[n_act] = (unsigned char)i;
              if (use_prev) step.use_prev_mask |= 1u << n_act;
              max_blocks = std::max(max_blocks, table[i].blocks);
              sl.run.ticket = SweepTicket{};
              sl.run.ticket.seq = step.seq[n_act];
              sl.run.t_issue = std::chrono::steady_clock::now();
              sl.wants = sl.wants_fit = false;
              sl.waiting = true;
              w->call_sweeps += 1;
              w->prof.grid_launches += 1;
              w->prof.reduce_launches += 1;
              w->prof.grid_bytes += 16ull * ((uint64_t)w->src.n + (uint64_t)w->tgt.n) +
                                    (kind == 0 ? 136ull * (uint64_t)table[i].blocks : 8ull * (uint64_t)w->src.n);
              w->prof.reduce_bytes += kind == 0 ? 136ull * (uint64_t)table[i].blocks : 40ull * (uint64_t)w->src.n + 136;
              if (kind == 1) {  // the ungated sweep's set-up, as sweep_issue / nn_keys_grid do it for one pair
                sl.run.ticket.few_host = reinterpret_cast<volatile int*>(w->h_sums + 20);
                *sl.run.ticket.few_host = -1;
                sl.run.ticket.red_src = w->src.data();
                sl.run.ticket.red_n = (int)w->src.n;
                sl.run.ticket.T = step.T[n_act];
                sl.run.ticket.thr = FLT_MAX;
                if (hipMemsetAsync(table[i].unmatched_count, 0, sizeof(int), gstream) != hipSuccess) {
                  fail(w, ICPGPU_ERR_HIP, "lock-step batch: memset");
                  return failed(ICPGPU_ERR_HIP, sl.pair, w);
                }
              }
              ++n_act;
            }
            if (n_act == 0) continue;
            bt_steps += 1;
            // kernel timing, sampled like the single-pair path's: a launch's HIP-event time / its pairs = one pair's sweep
            auto take_timing = [&](bool wait) {
              if (!timed_pairs) return;
              if (wait) (void)hipEventSynchronize(lead->ev[3]);
              else if (hipEventQuery(lead->ev[3]) != hipSuccess) return;
              float ms = 0.f;
              if (hipEventElapsedTime(&ms, lead->ev[2], lead->ev[3]) == hipSuccess) {
                lead->prof.grid_ms += (double)ms;  // (summed over the launch's pairs: grid_ms / grid_timed stays "per pair and sweep")
                lead->prof.grid_timed += (uint64_t)timed_pairs;
              }
              timed_pairs = 0;
            };
            take_timing(false);
            const bool timed = kind == 0 && timed_pairs == 0 && (step_counter++ % 5u) == 0;
            if (timed) (void)hipEventRecord(lead->ev[2], gstream);
            const BatchPair* d_table = static_cast<const BatchPair*>(lead->batch_table.ptr);
            hipError_t e = launch_nn_grid_search_batch(d_table, step, n_act, max_blocks, group_pack, kind == 1, gstream);
            if (timed) {
              (void)hipEventRecord(lead->ev[3], gstream);
              timed_pairs = n_act;
            }
            if (e == hipSuccess && kind == 0) e = launch_reduce_final_batch(d_table, step, n_act, gstream);
            if (e != hipSuccess) {
              fail(lead, ICPGPU_ERR_HIP, "lock-step batch launch: %s", hipGetErrorString(e));
              return failed(ICPGPU_ERR_HIP, slots[step.slot[0]].pair, lead);
            }
            if (kind == 1) {
              // per pair: the few points the grid left unmatched (completed on the device), then the keys-path reduction into
              // the pair's mailbox -- three small launches each, queued without waiting
              for (int a2 = 0; a2 < n_act; ++a2) {
                Slot& sl = slots[step.slot[a2]];
                icpgpu_ctx* w = sl.w;
                const BatchPair& bp = table[step.slot[a2]];
                hipError_t e2 = launch_nn_brute_few(w->src.data(), bp.unmatched, bp.unmatched_count, 0, w->tgt.data(), (int)w->tgt.n,
                                                    step.T[a2], bp.keys, reinterpret_cast<int*>(w->h_sums_dev + 20), gstream);
                int rc2 = e2 == hipSuccess ? ensure(w, w->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double)) : ICPGPU_ERR_HIP;
                if (!rc2 && launch_reduce(w->src.data(), (int)w->src.n, w->tgt.data(), bp.keys, step.T[a2], FLT_MAX,
                                          static_cast<double*>(w->partials.ptr), w->h_sums_dev, w->h_flags_dev, step.seq[a2], gstream) != hipSuccess)
                  rc2 = ICPGPU_ERR_HIP;
                if (rc2) {
                  fail(w, rc2, "lock-step batch: fitness sweep");
                  return failed(rc2, sl.pair, w);
                }
              }
            }
          }
          // wait for something to come back, then take everything that has
          bool progressed = false;
          for (size_t i = 0; i < n_slots; ++i) {
            Slot& sl = slots[i];
            if (sl.run.phase != P2PRun::Iterating && sl.run.phase != P2PRun::Fitness) continue;
            if (sl.lock && !sl.waiting) continue;  // its sweep has not been launched yet
            if (!sweep_ready(sl.w, sl.run.ticket)) continue;
            sl.waiting = false;
            int deferred = 0;
            const int rc = p2p_advance(sl.w, sl.run, (sl.lock && sl.run.phase == P2PRun::Iterating) ? &deferred : nullptr);
            if (rc) return failed(rc, sl.pair, sl.w);
            sl.wants = deferred == 1;
            sl.wants_fit = deferred == 2;
            if (sl.run.phase == P2PRun::Done) --live;
            progressed = true;
          }
          if (progressed) {
            idle_spins = 0;
            continue;
          }
          if ((++idle_spins & 0x3FFu) == 0) {  // nothing moved for a while: a faulted or hung kernel must not keep us here
            const auto now = std::chrono::steady_clock::now();
            const hipError_t q = hipStreamQuery(gstream);
            for (size_t i = 0; i < n_slots; ++i) {
              Slot& sl = slots[i];
              if (sl.run.phase != P2PRun::Iterating && sl.run.phase != P2PRun::Fitness) continue;
              if (q != hipSuccess && q != hipErrorNotReady) {
                fail(sl.w, ICPGPU_ERR_HIP, "HIP error while waiting for a reduction: %s", hipGetErrorString(q));
                return failed(ICPGPU_ERR_HIP, sl.pair, sl.w);
              }
              if (std::chrono::duration<double, std::milli>(now - sl.run.t_issue).count() > timeout_ms) {
                fail(sl.w, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for a kernel's result (hung kernel?)", timeout_ms);
                return failed(ICPGPU_ERR_HIP, sl.pair, sl.w);
              }
            }
          }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
        if (timed_pairs) {  // (the group is finished: its last timed launch is, too)
          float ms = 0.f;
          if (hipEventSynchronize(lead->ev[3]) == hipSuccess && hipEventElapsedTime(&ms, lead->ev[2], lead->ev[3]) == hipSuccess) {
            lead->prof.grid_ms += (double)ms;
            lead->prof.grid_timed += (uint64_t)timed_pairs;
          }
          timed_pairs = 0;
        }
        if (bt_on) {
          const auto bt3 = std::chrono::steady_clock::now();
          bt_fill += std::chrono::duration<double, std::milli>(bt1 - bt0).count();
          bt_build += std::chrono::duration<double, std::milli>(bt2 - bt1).count();
          bt_iter += std::chrono::duration<double, std::milli>(bt3 - bt2).count();
          bt_groups += 1;
        }
      }
    }
    std::vector<P2PRun> runs(depth);
    std::vector<size_t> pair_of(depth, 0);
    bool exhausted = false;
    for (unsigned idle_spins = 0;;) {
      bool progressed = false;
      size_t in_flight = 0;
      for (size_t s = 0; s < depth; ++s) {
        P2PRun& r = runs[s];
        icpgpu_ctx* w = ws[s];
        if (r.phase == P2PRun::Idle || r.phase == P2PRun::Done) {
          if (exhausted || abort.load()) continue;
          const size_t k = next.fetch_add(1);
          if (k >= n_pairs) {
            exhausted = true;
            continue;
          }
          pair_of[s] = k;
          int rc = load_pair(w, k);
          if (!rc) rc = p2p_begin(w, r, nullptr, nullptr, want_fitness, &results[k]);
          if (rc) return failed(rc, k, w);
          progressed = true;
          if (r.phase != P2PRun::Done) ++in_flight;
        } else if (sweep_ready(w, r.ticket)) {
          const int rc = p2p_advance(w, r);
          if (rc) return failed(rc, pair_of[s], w);
          progressed = true;
          if (r.phase != P2PRun::Done) ++in_flight;
        } else {
          ++in_flight;
        }
      }
      if (in_flight == 0 && (exhausted || abort.load())) return;
      if (progressed) {
        idle_spins = 0;
        continue;
      }
      if ((++idle_spins & 0x3FFu) == 0) {  // nothing moved for a while: a faulted or hung kernel must not keep us here
        const auto now = std::chrono::steady_clock::now();
        for (size_t s = 0; s < depth; ++s) {
          P2PRun& r = runs[s];
          if (r.phase != P2PRun::Iterating && r.phase != P2PRun::Fitness) continue;
          const hipError_t q = hipStreamQuery(ws[s]->stream);
          if (q != hipSuccess && q != hipErrorNotReady) {
            fail(ws[s], ICPGPU_ERR_HIP, "HIP error while waiting for a reduction: %s", hipGetErrorString(q));
            return failed(ICPGPU_ERR_HIP, pair_of[s], ws[s]);
          }
          if (std::chrono::duration<double, std::milli>(now - r.t_issue).count() > timeout_ms) {
            fail(ws[s], ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for a kernel's result (hung kernel?)", timeout_ms);
            return failed(ICPGPU_ERR_HIP, pair_of[s], ws[s]);
          }
        }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  };
  std::vector<std::thread> threads;
  for (size_t t = 1; t < n_threads; ++t) threads.emplace_back(work, t);
  work(0);
  for (auto& th : threads) th.join();
  for (size_t i = 0; i < n_ctx; ++i) {  // fold the workers' kernel accounting into the parent's profile
    icpgpu_profile& p = c->workers[i]->prof;
    c->prof.nn_launches += p.nn_launches; c->prof.nn_ms += p.nn_ms; c->prof.nn_pairs += p.nn_pairs; c->prof.nn_bytes += p.nn_bytes;
    c->prof.reduce_launches += p.reduce_launches; c->prof.reduce_ms += p.reduce_ms; c->prof.reduce_bytes += p.reduce_bytes;
    c->prof.transform_launches += p.transform_launches; c->prof.transform_ms += p.transform_ms; c->prof.transform_bytes += p.transform_bytes;
    c->prof.iterations += p.iterations; c->prof.aligns += p.aligns;
    c->prof.grid_launches += p.grid_launches; c->prof.grid_ms += p.grid_ms; c->prof.grid_bytes += p.grid_bytes;
    c->prof.nn_timed += p.nn_timed; c->prof.grid_timed += p.grid_timed; c->prof.reduce_timed += p.reduce_timed;
    c->prof.grid_bounded += p.grid_bounded;
    c->prof.grid_builds += p.grid_builds; c->prof.grid_build_ms += p.grid_build_ms; c->prof.grid_fallback_points += p.grid_fallback_points;
    c->prof.voxel_launches += p.voxel_launches; c->prof.voxel_ms += p.voxel_ms; c->prof.voxel_bytes += p.voxel_bytes;
    c->prof.gicp_cov_launches += p.gicp_cov_launches; c->prof.gicp_cov_ms += p.gicp_cov_ms; c->prof.gicp_cost_launches += p.gicp_cost_launches;
    c->prof.gicp_eval_ms += p.gicp_eval_ms; c->prof.gicp_eval_corr += p.gicp_eval_corr; c->prof.gicp_cov_points += p.gicp_cov_points;
    c->prof.targets_recognised += p.targets_recognised;
    c->prof.brute_bound_violations += p.brute_bound_violations;
    if (p.brute_bound_worst > c->prof.brute_bound_worst) c->prof.brute_bound_worst = p.brute_bound_worst;
    std::memset(&p, 0, sizeof(p));
  }
  for (const ThreadError& e : errors)  // the first failure in thread order (each thread stops at its first)
    if (e.code != ICPGPU_OK) {
      c->err = "align_batch pair " + std::to_string(e.pair) + ": " + e.msg;
      return e.code;
    }
  return ICPGPU_OK;
}

int icpgpu_nn(icpgpu_ctx* c, const float* T, int32_t* idx, float* d2) {
  ENTER(c);
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "nn: source and target must be set first");
  const int n_s = (int)c->src.n, n_t = (int)c->tgt.n;
  if (n_s == 0) return ICPGPU_OK;
  if (!idx || !d2) return fail(c, ICPGPU_ERR_INVALID_ARG, "null output");
  int rc = ensure(c, c->keys, (size_t)n_s * sizeof(unsigned long long));
  if (rc) return rc;
  if ((rc = ensure(c, c->idx, (size_t)n_s * sizeof(int32_t)))) return rc;
  if ((rc = ensure(c, c->d2, (size_t)n_s * sizeof(float)))) return rc;
  if ((rc = ensure_grid(c, threshold_from(c->params.max_correspondence_distance * c->params.max_correspondence_distance))))
    return rc;
  auto* keys = static_cast<unsigned long long*>(c->keys.ptr);
  const Xform X = to_xform(T);
  const bool use_grid = grid_ready(c);
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (use_grid) {
    if ((rc = nn_keys_grid(c, X, keys))) return rc;
  } else {
    if ((rc = nn_keys_brute(c, c->tgt.data(), n_t, X, keys))) return rc;
  }
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(c, launch_unpack_keys(keys, n_s, static_cast<int32_t*>(c->idx.ptr), static_cast<float*>(c->d2.ptr), c->stream));
  HIP_TRY(c, hipMemcpyAsync(idx, c->idx.ptr, (size_t)n_s * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d2, c->d2.ptr, (size_t)n_s * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  if (use_grid) {
    c->prof.grid_launches += 1;
    c->prof.grid_ms += ms;
    c->prof.grid_bytes += 16ull * ((uint64_t)n_s + (uint64_t)n_t) + 8ull * (uint64_t)n_s;
  } else {
    c->prof.nn_launches += 1;
    c->prof.nn_ms += ms;
    c->prof.nn_pairs += (uint64_t)n_s * (uint64_t)n_t;
    c->prof.nn_bytes += 16ull * ((uint64_t)n_s + (uint64_t)n_t) + 8ull * (uint64_t)n_s;
  }
  return ICPGPU_OK;
}

int icpgpu_reduce(icpgpu_ctx* c, const float* T, double max_dist, double sums[17]) {
  ENTER(c);
  if (!sums) return fail(c, ICPGPU_ERR_INVALID_ARG, "sums is null");
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "reduce: source and target must be set first");
  if (!c->keys.ptr || c->keys.cap < c->src.n * sizeof(unsigned long long))
    return fail(c, ICPGPU_ERR_NO_INPUT, "reduce: no nearest-neighbour sweep to reduce (call icpgpu_nn first)");
  {
    int rc = ensure(c, c->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double));
    if (rc) return rc;
  }
  HIP_TRY(c, launch_reduce(c->src.data(), (int)c->src.n, c->tgt.data(), static_cast<unsigned long long*>(c->keys.ptr),
                           to_xform(T), threshold_from(max_dist * max_dist), static_cast<double*>(c->partials.ptr),
                           static_cast<double*>(c->sums.ptr), nullptr, 0, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_sums, c->sums.ptr, kReduceTerms * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::memcpy(sums, c->h_sums, kReduceTerms * sizeof(double));
  return ICPGPU_OK;
}

int icpgpu_solve(const double sums[17], double Tk[16]) {
  if (!sums || !Tk) return ICPGPU_ERR_INVALID_ARG;
  Mat4d M;
  const bool ok = solve_umeyama(sums, M);
  for (int i = 0; i < 16; ++i) Tk[i] = M[i];
  return ok ? ICPGPU_OK : ICPGPU_ERR_INVALID_ARG;
}

int icpgpu_transform(icpgpu_ctx* c, const float* T, float* out_xyzw) {
  ENTER(c);
  if (!c->src.set) return fail(c, ICPGPU_ERR_NO_INPUT, "transform: no source set");
  if (c->src.n && !out_xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "out is null");
  return write_output_cloud(c, to_xform(T), out_xyzw);
}

int icpgpu_voxel_grid(icpgpu_ctx* c, const float* xyzw, size_t n, float leaf, float* out_xyzw, size_t* n_out) {
  ENTER(c);
  if (!n_out || (n && !xyzw)) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (n > (size_t)INT32_MAX - 4096) return fail(c, ICPGPU_ERR_INVALID_ARG, "cloud too large: %zu points", n);
  *n_out = 0;
  c->vox_last_n = 0;
  int rc = ensure(c, c->vox_in, n * sizeof(float4));
  if (rc) return rc;
  if (n) HIP_TRY(c, hipMemcpyAsync(c->vox_in.ptr, xyzw, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  int m = 0;
  bool pass = false;
  if ((rc = voxel_filter_device(c, static_cast<const float4*>(c->vox_in.ptr), (int)n, leaf, c->vox_out, &m, &pass))) return rc;
  if (m && out_xyzw) {
    HIP_TRY(c, hipMemcpyAsync(out_xyzw, c->vox_out.ptr, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->vox_last_n = (size_t)m;  // out_xyzw == NULL: the filtered cloud waits in HBM for icpgpu_voxel_grid_fetch
  *n_out = (size_t)m;
  return ICPGPU_OK;
}

int icpgpu_voxel_grid_fetch(icpgpu_ctx* c, float* out_xyzw, size_t capacity, size_t* n_out) {
  ENTER(c);
  const size_t m = c->vox_last_n;
  if (n_out) *n_out = m;
  if (m > capacity) return fail(c, ICPGPU_ERR_INVALID_ARG, "voxel_grid_fetch: %zu points, room for %zu", m, capacity);
  if (m && !out_xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (m) {
    HIP_TRY(c, hipMemcpyAsync(out_xyzw, c->vox_out.ptr, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
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
  if ((rc = voxel_filter_device(c, static_cast<const float4*>(c->vox_in.ptr), (int)n, leaf, c->src.buf, &m, &pass))) return rc;
  c->src.n = (size_t)m;
  c->src.set = true;
  if (n_out) *n_out = (size_t)m;
  return ICPGPU_OK;
}

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
  return ICPGPU_OK;
}

// ---- the mapper's map (octree_mapper.cpp:55-90,133-172) ------------------------------------------
namespace {

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

// The lattice is anchored by the first point that is ever added: origin = p - resolution / 2 (PCL
// OctreePointCloud::adoptBoundingBoxToPoint for an empty octree).  d_in: the batch in HBM.
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
        M.desc.ox = (double)p[0] - M.desc.res / 2.0;
        M.desc.oy = (double)p[1] - M.desc.res / 2.0;
        M.desc.oz = (double)p[2] - M.desc.res / 2.0;
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

}  // namespace

namespace {
// OctreePointCloud::adoptBoundingBoxToPoint for one point outside the box (oracle/map_approx_np.py::_adopt restates the same):
// the box doubles -- all three axes at once, the old root becoming the UPPER child on every axis the point does not violate
// from above -- until the point is inside
void approx_adopt(VoxelMap& M, const float p[3]) {
  ApproxBox& b = M.box;
  const double eps = (double)FLT_EPSILON;
  for (;;) {
    if (!M.box_defined) {
      for (int a = 0; a < 3; ++a) {
        b.min[a] = (double)p[a] - b.res / 2.0;
        b.max[a] = (double)p[a] + b.res / 2.0;
      }
      b.depth = 0;
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
      if (!upper[a]) b.min[a] -= side;
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
    M.box_upto += 1;
    if (M.box.depth > approx_max_depth())
      return fail(c, ICPGPU_ERR_UNSUPPORTED, "map: the octree would be %d levels deep (approximate search supports %d)", M.box.depth,
                  approx_max_depth());
  }
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
  HIP_TRY(c, launch_approx_insert(M.pts.data(), M.nodes_upto, M.n, M.box, nk, nv, M.node_cap, c->stream));
  M.nodes_upto = M.n;
  // (3) the descent
  HIP_TRY(c, launch_approx_descend(c->src.data(), n_s, T, M.box, nk, nv, M.node_cap, keys, c->stream));
  return ICPGPU_OK;
}
}  // namespace

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
    HIP_TRY(c, hipMemcpyAsync(nn_out_xyzw, c->tgt.buf.ptr, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (n_nn) *n_nn = (size_t)m;
  return ICPGPU_OK;
}

int icpgpu_profile_reset(icpgpu_ctx* c) {
  if (!c) return ICPGPU_ERR_INVALID_ARG;
  (void)resolve_sweep_timings(c);
  std::memset(&c->prof, 0, sizeof(c->prof));
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
  *out = c->prof;
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
