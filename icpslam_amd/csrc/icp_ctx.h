// icp_ctx.h -- what the units of the C-ABI share (INTERNAL: nothing here is part of include/icpgpu.h): the context, its
// buffers and caches, and the helpers that cross unit boundaries.  The C-ABI of libicpgpu.so is implemented by
//   icpgpu_context.cpp  create / destroy, parameters, clouds (upload, recognition, promote), profile, stream
//   icpgpu_index.cpp    the target's uniform grid: resumable builds, cell-size rules, the grid search's dispatch
//   icpgpu_p2p.cpp      point-to-point ICP: sweeps (search + fused reduction), the mailbox, the resumable run, align / fitness
//   icpgpu_gicp.cpp     GICP: covariances, the evaluation server, the BFGS outer loop
//   icpgpu_batch.cpp    icpgpu_align_batch: lock-step groups and the round-robin scheduler
//   icpgpu_voxel.cpp    the voxel-grid filter's host side
//   icpgpu_map.cpp      the mapper's map and its nn cloud
// (until round 4 all of this was one 3 200-line icpgpu_api.cpp).
#pragma once

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
#include "icp_env.h"

using namespace icpgpu;

struct icpgpu_ctx;

namespace icpgpu_impl {



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
  // a box that CONTAINS every finite point (encoded like bbox_kernel's result), valid while the owner's version counter equals
  // bbox_version (0 = none): what a grid build needs of the cloud before its count pass.  Filled by the voxel filter (the raw
  // scan's box contains its centroids) and by a grid build's own bounding-box pass; mutable: a cache, not content.
  mutable uint64_t bbox_version = 0;
  mutable int bbox_enc[6] = {0, 0, 0, 0, 0, 0};
  mutable bool bbox_exact = false;  // from a bounding-box pass over these very points (no containment check needed)
  const float4* data() const { return static_cast<const float4*>(buf.ptr); }
};

// Uniform grid over the target (icp_grid.hip), built lazily for the cutoff of the current parameters.
struct GridIndex {
  bool built = false, usable = false;
  uint64_t version = 0;      // target version it was built for
  float cutoff = 0.f;        // correspondence distance it was built for
  double cut_d = 0.0;        // the same before rounding (r_max is derived from this one)
  bool adopted = false;      // c->grid only: this is the grid the target's GICP covariances were computed over (ensure_grid)
  GridDesc g{};
  int n_binned = 0, max_pop = 0;
  double point_population = 0.0;  // cell population seen by a random point (sum c^2 / sum c)
  DeviceBuf sorted, cell_start, cell_of_point, rank, block_sums, ints, unmatched, leftover;
  uint64_t serial = 0;       // unique per completed build (grids change hands between c->grid and c->src_grid)
};

// The neighbour every query found in the last grid sweep (nn_quad_kernel), as its position in the grid's sorted copy (4 bytes
// per query): an upper bound for the next sweep of the same queries over the same build of the same grid, whatever the
// transform (icp_grid.hip).  `valid` is cleared at the start of every
// alignment, so an alignment never depends on the one before it.
struct PrevNeighbours {
  DeviceBuf buf;
  bool valid = false;
  const void* src = nullptr;     // query array the entries are indexed by
  const void* sorted = nullptr;  // grid they were found in
  int n = 0;
  int n_binned = 0;              // points in that grid's sorted copy (the positions stay below 16 x this)
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
  ApproxHistory box_hist{};   // every version of the box with the map index it is in force from (PCL keys are never recomputed)
  long long box_shift[3] = {0, 0, 0};  // voxels the minimum has moved since the first box
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
constexpr int kMfmaMinPoints = 8192;             // the matrix-core brute-force kernels from this many points on (both clouds)
constexpr size_t kGridMinTarget = 4096;         // AUTO: below this the brute-force kernel is launch-latency bound anyway


}  // namespace icpgpu_impl
using namespace icpgpu_impl;

constexpr size_t kMaxServerWorkers = 8;

// ICPGPU_GICP_DEVICE: where GICP's inner BFGS runs -- 0: on the host over the evaluation server; 1: inside a resident kernel
// (icp_gicp.hip: gicp_solve_kernel); unset or "auto": MEASURED per context.  Same bits either way (tests/test_gpu_gicp.py,
// 1 200 campaign registrations through both).  Which one is faster depends on the box: the host loop's evaluation is 6.8-8.3 us
// with the box's PCIe and CPU, the kernel's 7.2-7.7 us wherever it runs (EXPERIMENTS.md section 9-f1) -- so a context times a few
// outer iterations each way (align_gicp) and keeps the faster; runs the one-XCD variant cannot take stay on the host.
// the shortest structs a caller may hand over (icpgpu_create_abi): the layouts of icpgpu.h 0.4, of which 1.0's are extensions (1.0
// added icpgpu_result.gicp_solver; the floor sits one step below the first sized release so that the min-copy rule has a shorter
// caller to be tested with: tests/test_gpu_errors.py)
constexpr size_t kAbiParams10 = 56, kAbiResult10 = 112, kAbiProfile10 = 352;
static_assert(sizeof(icpgpu_params) >= kAbiParams10 && sizeof(icpgpu_result) >= kAbiResult10 && sizeof(icpgpu_profile) >= kAbiProfile10,
              "public structs only grow (include/icpgpu.h, ABI rule)");
inline int gicp_device_solver_mode() {  // 0 host, 1 device, 2 measured
  static const int v = [] {
    const char* e = std::getenv("ICPGPU_GICP_DEVICE");
    if (!e || std::strcmp(e, "auto") == 0) return 2;
    return std::atoi(e) != 0 ? 1 : 0;
  }();
  return v;
}
inline bool gicp_device_solver_enabled() { return gicp_device_solver_mode() != 0; }  // (its buffers are allocated with the context)
inline bool gicp_server_enabled() {  // ICPGPU_GICP_SERVER=0: every GICP evaluation is its own launch
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
  bool cov_timing_pending = false;  // ev[2] .. ev[3] bracket a covariance pass whose duration has not been read yet (resolve_cov_timing)
  double cov_h_hint = 0.0, cov_h_hint_cut = 0.0;  // the cell size the last covariance grid settled on (ensure_covariances)
  // A covariance grid built WITHOUT waiting for its occupancy statistics (icpgpu_gicp.cpp: ensure_covariances, speculative form):
  // the statistics were posted into the mailbox (pairs 32..) behind the count pass and are looked at when the alignment first
  // waits for the device anyway; `cooldown` alignments after one that failed the check go the waiting way.
  struct SpecGrid {
    bool pending = false;
    unsigned long long number = 0;  // the post's sequence number
    uint64_t serial = 0;            // GridIndex::serial of the build
    int n = 0;                      // points the cloud holds: all of them must have been binned
    double h = 0.0, knn_population = 0.0, cut = 0.0;
  } spec_grid;
  int spec_cooldown = 0;
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
  bool mailbox_release = false;  // result pairs leave the device as value / system-scope release / tag (icp_kernels.h)
  // GICP cost evaluations: per-workgroup partials (kGicpDirectBlocks x 17) + one flag per workgroup, same kind of memory
  double* h_gicp = nullptr;
  double* h_gicp_dev = nullptr;
  volatile unsigned long long* h_gicp_flags = nullptr;
  unsigned long long* h_gicp_flags_dev = nullptr;
  // resident evaluation server (icp_gicp.hip): its command line, fine-grained device memory the host writes through the BAR
  unsigned int* gicp_cmd = nullptr;
  // the device solver (icp_gicp.hip: gicp_solve_kernel): the workgroups' partial-sum slots (fine-grained device memory), the
  // result granules (host, mapped), the number the next run's granules carry, and whether the path is (still) in use
  unsigned long long* gicp_slots = nullptr;
  volatile unsigned long long* h_solve = nullptr;
  unsigned long long* h_solve_dev = nullptr;
  unsigned long long gicp_solve_seq = 0;
  bool gicp_device_ok = false;
  int gicp_device_failures = 0;  // batch runs in a row whose device solver gave no answer (three switch it off on this worker)
  bool gicp_resources_ready = false;  // ensure_gicp_resources has run (icpgpu_context.cpp)
  // the quadratic inner solver (icpgpu_params.gicp_inner; icp_gicp_quadratic.h): the workgroups' partial sums, the counter of
  // finished workgroups, the 2 x kGicpQuadSums result pairs (host, mapped) and the number the next pass's pairs carry
  double* quad_partials = nullptr;
  unsigned int* quad_done = nullptr;
  volatile unsigned long long* h_quad = nullptr;
  unsigned long long* h_quad_dev = nullptr;
  unsigned long long quad_seq = 0;
  bool quad_pending = false;  // a pass was launched and its sums have not been read: the next launch zeroes the counter first (a pass that
                              // died half-way would leave it mid-count)
  // measured mode: 0 = still timing both solvers, 1 = host, 2 = device; microseconds and evaluations of the timed inner
  // minimisations, [0] host [1] device (the first run of each is a warm-up and not counted)
  int gicp_choice = 0;            // (set at creation since 1.0: 1 host / 2 device; 0 only while icpgpu_calibrate is timing both)
  // what the CALLER's header says the three public structs measure (icpgpu_create_abi; include/icpgpu.h "ABI rule")
  size_t abi_params = sizeof(icpgpu_params), abi_result = sizeof(icpgpu_result), abi_profile = sizeof(icpgpu_profile);
  double gicp_cal_us[2] = {0.0, 0.0};
  unsigned long long gicp_cal_evals[2] = {0, 0};
  unsigned int gicp_cal_runs[2] = {0, 0};
  // ... its one-XCD variant: granule slots in ordinary device memory (the XCD's L2 is the medium), the worker claims, the XCD
  unsigned long long* gicp_slots_local = nullptr;
  unsigned long long* gicp_owner = nullptr;
  int gicp_xcc = 0;
  bool gicp_local_ok = false;
  bool gicp_server_on = false;
  bool gicp_server_allowed = true;  // align_batch with more than kMaxServerWorkers threads: single launches (below)
  int gicp_blocks_most = kGicpDirectBlocks;
  // ICPGPU_GICP_TIMING=1 (development): where an evaluation's microseconds go, printed when the context is destroyed
  double gt_cmd = 0, gt_wait = 0, gt_merge = 0, gt_between = 0, gt_dev_wait = 0, gt_dev_work = 0;
  unsigned long long gt_n = 0;
  double gt_dev_reduce = 0, gt_trickle = 0;
  unsigned long long gt_dev_n = 0;
  int gt_pending = 0;
  double gt_stage[10] = {};  // ICPGPU_GICP_TIMING: host wall per stage of align_gicp (icpgpu_gicp.cpp), microseconds
  unsigned long long gt_aligns = 0;
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
  void* h_stage = nullptr;   // pinned, mapped staging for results that go to the caller's pageable buffers (copy_to_host; round 6:
  void* h_stage_dev = nullptr;  // kernels write result clouds straight into it -- output_cloud_issue, icpgpu_voxel_grid_view)
  size_t h_stage_cap = 0;
  bool want_view = false;       // the entry point in progress hands the staged cloud out as a view (no copy to a caller's buffer)
  volatile unsigned long long* h_post = nullptr;  // mapped, coherent: 32 result pairs for fetch_ints, 8 for a grid's statistics
                                                  // (spec_grid), 8 for a staged cloud's marker (stage_post) -- icpgpu_context.cpp
  unsigned long long* h_post_dev = nullptr;
  unsigned long long post_seq = 0;
  bool have_final = false;
  Mat4d final_T = mat4_identity();
  icpgpu_profile prof{};
  int nn_variant = -1;  // ICPGPU_NN_VARIANT: a variant of the plain-VALU brute-force kernel (-1: none forced)
  DeviceBuf cov_list;  // GICP covariances: the points the selecting kernel leaves to the far-field and the streaming kernel (counts + indices)
  void* cov_list_zeroed = nullptr;  // the allocation (address, size) whose two counters are known to be zero (the finish kernel leaves them so)
  size_t cov_list_zeroed_cap = 0;
  DeviceBuf vox_in, vox_out, vox_keys, vox_vals, vox_flags, vox_slots, vox_temp, vox_ints;  // voxel filter scratch
  DeviceBuf vox_bins, vox_pub;   // ... of the direct (no library sort) path: self-cleaning histogram + group ranges; published counts
  int* vox_bbox_ready = nullptr; // vox_ints' address when its bounding-box slots are known to hold the initial values (icpgpu_voxel.cpp)
  size_t vox_last_n = 0;         // points of the last icpgpu_voxel_grid result (still in vox_out)
  // ... what icpgpu_set_source needs to recognise that result when the caller hands it back as a host buffer (the reference's
  // voxelFilterCloud -> setInputSource sequence, icp_odometer.cpp:177,193): its content fingerprint (taken on the device while
  // the fetch's copy is in flight), the fingerprint of its 256 sampled points (from the fetched host copy), and the raw scan's
  // bounding box, which contains every centroid (what a grid build needs of the cloud before its count pass)
  bool vox_fp_valid = false;
  unsigned long long vox_fp = 0, vox_sample_fp = 0;
  bool vox_box_valid = false;
  int vox_box[6] = {0, 0, 0, 0, 0, 0};
  void* vox_bins_zeroed = nullptr;  // the allocation (address, size) whose histogram is known to be zero
  size_t vox_bins_zeroed_cap = 0;
  void* vox_pub_zeroed = nullptr;
  size_t vox_pub_zeroed_cap = 0;
  std::vector<icpgpu_ctx*> workers;  // align_batch: one sub-context (own stream + scratch) per host worker thread
  DeviceBuf batch_table;             // lock-step batch: the BatchPair table of the group this context leads
  std::atomic<size_t> batch_table_cells{0};   // align_batch: the largest cell table any worker has needed (icpgpu_index.cpp)
  std::atomic<size_t>* shared_table_cells = nullptr;  // a batch worker: its parent's batch_table_cells
  size_t batch_presized_src = 0, batch_presized_tgt = 0, batch_presized_cells = 0;  // a batch worker: what its buffers were sized for up front
  std::vector<hipStream_t> group_streams;  // lock-step batch: one stream per group, created consecutively (icpgpu_batch.cpp)
  int host_share = 1;                // batch drivers of this process that share its CPUs with this context (icp_multi.cpp)
  std::string err;
};

// the number a kernel is handed for its result pairs: the plain number, marked when the mailbox is in release mode
inline unsigned long long wire_seq(const icpgpu_ctx* c, unsigned long long seq) { return c->mailbox_release ? (seq | kMailboxReleaseBit) : seq; }

#define HIP_TRY(c, expr)                                                                              \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail((c), e_ == hipErrorOutOfMemory ? ICPGPU_ERR_OOM : ICPGPU_ERR_HIP, "%s failed: %s", #expr, \
                  hipGetErrorString(e_));                                                             \
  } while (0)

#define ENTER(c)                                                   \
  if (!(c)) return fail(nullptr, ICPGPU_ERR_INVALID_ARG, "null context"); \
  HIP_TRY((c), hipSetDevice((c)->device))

namespace icpgpu_impl {

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
  bool post = false;        // build_grid: the read-backs go through fetch_ints (the batch scheduler copies and synchronises its groups itself)
  bool box_cached = false;  // the bounding box came from the cloud's cache (a CONTAINING box: gb_on_count checks that every point was binned)
  double h_start = 0.0;  // > 0: the first count pass uses this cell size instead of cut / 4.5 (k-NN grids only: see ensure_covariances)
  double h = 0.0;
  int attempt = 0;
  bool shrunk = false;
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  GridDesc g{};
  std::chrono::steady_clock::time_point t0{};
};

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

// A point-to-point alignment as a resumable run: begin() queues the first sweep, advance() -- called once the sweep's sums
// have arrived -- does the host's share of an iteration (Umeyama / SVD, convergence test: icp_solver.cpp) and queues the
// next sweep, the fitness sweep, or finishes.  icpgpu_align drives one run to the end; icpgpu_align_batch keeps several
// contexts' runs in flight from one host thread.
struct StageTicket {  // a cloud on its way into the pinned staging buffer (stage_post / stage_wait below)
  unsigned long long number = 0;
  int n_ints = 0;
  bool issued = false;
};

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
  StageTicket out_ticket;  // the aligned cloud, queued in front of the fitness sweep (p2p_advance), taken out while the sweep runs
  bool out_done = false;   // ... it has been (align_p2p) -- p2p_finish has nothing left to write
  std::chrono::steady_clock::time_point t_start, t_issue;
};

// A GICP registration as a resumable run (icpgpu_gicp.cpp: gicp_run_begin / gicp_run_step): the counterpart of P2PRun for the
// solver the reference instantiates, so that one host thread of icpgpu_align_batch keeps several registrations in flight.
struct GicpRun {
  enum Phase { Idle, Blocking, CovGrid, WantSolve, Solve, Quad, Fitness, Done } phase = Idle;  // Quad: waiting for the quadratic form's sums
  bool quadratic = false;  // icpgpu_params.gicp_inner = QUADRATIC: every outer iteration is search + one pass + BFGS on the host
  // combine: the run does not launch its outer iteration's solver itself -- it stops in WantSolve with `item` filled in, and the
  // batch scheduler launches the solvers of all the runs that are ready in ONE kernel (gicp_solve_batch_kernel) on solve_stream
  bool combine = false;
  GicpSolveItem item{};
  hipStream_t solve_stream = nullptr;  // where the run's solver was launched (its own stream unless combined)
  float guess[16], transformation[16], previous[16];
  float thr = 0.f, thr_excl = 0.f;
  int nr = 0, state = ICPGPU_NOT_CONVERGED, want_fitness = 0, cov_stage = 0;
  bool converged = false, local = false;
  unsigned n_corr = 0;
  double mse = 0.0;
  GridBuild gb;
  unsigned long long marker = 0, seq0 = 0;
  unsigned polls = 0;
  SweepTicket ticket;
  icpgpu_result* res = nullptr;
  std::chrono::steady_clock::time_point t_start, t_issue;
};

// ---- helpers that cross unit boundaries -------------------------------------------------------------------------------------
// icpgpu_context.cpp
int fail(icpgpu_ctx* c, int code, const char* fmt, ...);
int ensure(icpgpu_ctx* c, DeviceBuf& b, size_t bytes);
extern std::atomic<unsigned long long> g_alloc_calls, g_alloc_us;  // allocations through ensure() and their host microseconds
void release(DeviceBuf& b);
Xform to_xform(const Mat4d& T);
Xform to_xform(const float* T);
float threshold_from(double r2);
// n <= 32 ints from device memory, queued behind everything the stream holds, into host_dst -- returns when they are there
// (a posted kernel + a polled mailbox instead of hipMemcpyAsync + hipStreamSynchronize: icp_kernels.hip post_ints_kernel)
int fetch_ints(icpgpu_ctx* c, const int* d_src, int n, int* host_dst);
int wait_posted(icpgpu_ctx* c, const volatile unsigned long long* box, int n, unsigned long long number, int* out, const char* what);
unsigned long long sample_fingerprint(const float* xyzw, size_t n);
int copy_to_host(icpgpu_ctx* c, void* dst, const void* d_src, size_t bytes, const int* d_extra = nullptr, int n_extra = 0, int* extra_out = nullptr);
// The staging buffer as a target of kernels: at least `bytes` of pinned, mapped, coherent host memory (c->h_stage / h_stage_dev);
// ICPGPU_ERR_UNSUPPORTED above kStageMaxBytes unless `any_size` (views have nowhere else to go).
constexpr size_t kStageMaxBytes = 8u << 20;
int ensure_stage(icpgpu_ctx* c, size_t bytes, bool any_size = false);
// A marker behind whatever was queued to fill the staging buffer (<= 8 ints of device memory ride on it), and the wait for it:
// when stage_wait returns, everything the stream wrote to host memory before the marker is visible.  (StageTicket: above P2PRun)
int stage_post(icpgpu_ctx* c, const int* d_ints, int n_ints, StageTicket& tk);
int stage_wait(icpgpu_ctx* c, StageTicket& tk, int* ints_out);
int set_cloud_host(icpgpu_ctx* c, Cloud& cl, const float* xyzw, size_t n, bool sync = true);
int set_cloud_device(icpgpu_ctx* c, Cloud& cl, const void* d_xyzw, size_t n);
int ensure_gicp_resources(icpgpu_ctx* c);
int ensure_gicp_quadratic_resources(icpgpu_ctx* c);
bool gicp_inner_quadratic(const icpgpu_ctx* c);  // params.gicp_inner, or ICPGPU_GICP_INNER's override
int create_context(icpgpu_ctx** out_ctx, int device_id, bool with_stream);
int ensure_stream(icpgpu_ctx* c);
int promote_internal(icpgpu_ctx* c);
// icpgpu_index.cpp
int gb_begin(icpgpu_ctx* c, GridBuild& b, const Cloud& cloud, uint64_t version, double cut, bool adapt, GridIndex& G,
             const int* orig_index = nullptr, double knn_population = 0.0, double h_start = 0.0);
int gb_advance(icpgpu_ctx* c, GridBuild& b);
int gb_finish_unchecked(icpgpu_ctx* c, GridBuild& b);  // WaitCount -> Done without the statistics (the caller checks them later)
int build_grid(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, double cut, bool adapt, GridIndex& G,
               const int* orig_index = nullptr, double knn_population = 0.0, double h_start = 0.0);
int ensure_grid(icpgpu_ctx* c, float accept_thr);
int grid_flags(const GridIndex& G, bool src_in_cell_order);
int prev_neighbours(icpgpu_ctx* c, const GridIndex& G, const float4* src_pts, int n_q, int flags, unsigned int*& buf, bool& use);
int nn_keys_grid(icpgpu_ctx* c, GridIndex& G, const float4* src_pts, int n_s, const float4* tgt_pts, int n_t, const Xform& T,
                 unsigned long long* keys, int* deferred = nullptr);
int complete_deferred_keys(icpgpu_ctx* c, GridIndex& G, const float4* src_pts, int n_s, const float4* tgt_pts, int n_t,
                           const Xform& T, unsigned long long* keys, int n_un);
int source_order_mode();
int nn_keys_grid(icpgpu_ctx* c, const Xform& T, unsigned long long* keys);
bool grid_ready(const icpgpu_ctx* c);
int ensure_source_order(icpgpu_ctx* c, float accept_thr);
bool source_ordered(const icpgpu_ctx* c);
int source_in_morton_order(icpgpu_ctx* c);
// icpgpu_p2p.cpp
int nn_keys_brute(icpgpu_ctx* c, const float4* tgt_pts, int n_t, const Xform& T, unsigned long long* keys, bool* used_mfma = nullptr);
int resolve_sweep_timings(icpgpu_ctx* c, bool block = true);
int resolve_cov_timing(icpgpu_ctx* c);  // icpgpu_gicp.cpp
double wait_timeout_ms();
bool flags_ready(const volatile unsigned long long* pairs, int n_pairs, unsigned long long seq);
void take_sums(icpgpu_ctx* c);
int wait_flags(icpgpu_ctx* c, const volatile unsigned long long* flags, int n_flags, unsigned long long seq);
int wait_sums(icpgpu_ctx* c, unsigned long long seq);
int sweep_issue(icpgpu_ctx* c, const Xform& T, float thr, bool open_range, SweepTicket& tk);
bool sweep_ready(const icpgpu_ctx* c, const SweepTicket& tk);
int sweep_complete(icpgpu_ctx* c, SweepTicket& tk);
int nn_and_reduce(icpgpu_ctx* c, const Xform& T, float thr, bool open_range);
int write_output_cloud(icpgpu_ctx* c, const Xform& T, float* out_xyzw);
// ... in two halves (round 6): the transform kernel writes the aligned cloud into the pinned staging buffer and a marker follows;
// the host takes it out (one memcpy, or nothing for a view) when the marker is there -- callers queue the fitness sweep in between,
// so the copy runs while the sweep does.  output_cloud_issue leaves tk.issued false when the cloud does not fit the staging buffer
// (write_output_cloud's copy-engine path is the caller's way then).
int output_cloud_issue(icpgpu_ctx* c, const Xform& T, float* out_xyzw, StageTicket& tk);
int output_cloud_complete(icpgpu_ctx* c, StageTicket& tk, float* out_xyzw);
void init_result(icpgpu_result* r);
int p2p_finish(icpgpu_ctx* c, P2PRun& r);
int p2p_prepare(icpgpu_ctx* c, P2PRun& r, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res);
int p2p_begin(icpgpu_ctx* c, P2PRun& r, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res);
int p2p_advance(icpgpu_ctx* c, P2PRun& r, int* deferred = nullptr);
int align_p2p(icpgpu_ctx* c, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res);
// icpgpu_voxel.cpp
int voxel_filter_device(icpgpu_ctx* c, const float4* d_in, int n, float leaf, DeviceBuf& out, int* n_out, bool* passthrough,
                        int* bbox_enc_out = nullptr, bool publish = false, bool* published = nullptr, unsigned long long* fp_sum = nullptr);
// icpgpu_gicp.cpp
int ensure_covariances(icpgpu_ctx* c, const Cloud& cloud, uint64_t version, GridIndex& G, DeviceBuf& cov, uint64_t& cov_version, bool allow_unchecked = false);
int covariance_grid_check(icpgpu_ctx* c);
int align_gicp(icpgpu_ctx* c, const float* guess_in, float* out_xyzw, int want_fitness, icpgpu_result* res);
int gicp_run_begin(icpgpu_ctx* c, GicpRun& r, int want_fitness, icpgpu_result* res, bool combine = false);
int gicp_run_step(icpgpu_ctx* c, GicpRun& r);  // < 0 error, 0 nothing yet, 1 moved on (r.phase == GicpRun::Done: finished)
void gicp_run_solver_launched(icpgpu_ctx* c, GicpRun& r, hipStream_t solve_stream);  // WantSolve -> Solve (the scheduler launched r.item)

}  // namespace icpgpu_impl
