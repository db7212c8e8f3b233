// icp_kernels.h -- launch wrappers for the gfx950 kernels of the ICP hot path (internal).
//
// Rows of SURVEY.md §8(a) implemented here:
//   a2  nearest-neighbour correspondence search   -> launch_nn_brute()
//   a3  outlier rejection (d2 <= r^2 predicate)   -> fused into launch_reduce()
//   a4  centroid / cross-covariance reduction     -> launch_reduce()
//   a6  point-set transform                       -> fused into a2/a4 loads; launch_transform() for the output cloud
//   a9  fitness score                             -> a2 + launch_reduce() with an open threshold
// Reference call sites these replace: /root/reference/src/icpslam/icp_odometer.cpp:198-201,
// /root/reference/src/icpslam/octree_mapper.cpp:114-117 (the arithmetic itself lives in PCL).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpgpu {

// Rigid transform as 12 floats (row-major 3x4) so that it travels in SGPRs as a kernel argument.
// ---- the result mailbox's pairs --------------------------------------------------------------------------------------------
// Every value the host (or another workgroup) waits for travels as a 16-byte pair {value bits, tag}, tag = the number of the
// sweep / evaluation it belongs to and a 24-bit checksum of the value's bits.  The pair leaves the device as ONE 16-byte
// write-through store (icp_device.h: store_pair_system) and has never been seen half-written on gfx950; but the reader does
// not rely on that: it accepts a pair only if the tag carries the number it waits for AND the checksum of the bits it read --
// a torn observation (new tag, old value, or the other way round) fails the check and is simply read again.  So a pair is
// valid or recognisably not, in whatever order its halves become visible (until round 4 correctness rested on the pair
// never being seen torn: VERDICT r3).  ICPGPU_MAILBOX=release (or a torn pair seen by the start-up self-test) selects the
// slower classic form on the device side -- value, system-scope release fence, tag -- signalled to the kernels by
// kMailboxReleaseBit in the number they are handed; the reader is the same.
#if defined(__HIPCC__)
#define ICPGPU_HD __host__ __device__
#else
#define ICPGPU_HD
#endif
static constexpr unsigned long long kMailboxReleaseBit = 1ull << 62;
ICPGPU_HD inline unsigned long long mailbox_tag(unsigned long long seq, unsigned long long bits) {  // seq < 2^38
  return (seq << 24) | ((bits ^ (bits >> 24) ^ (bits >> 48)) & 0xFFFFFFull);
}
// host side: true and the value bits if the pair carries number seq and its own checksum
inline bool mailbox_read(const volatile unsigned long long* pair, unsigned long long seq, unsigned long long* bits_out) {
  const unsigned long long tag = pair[1], bits = pair[0];
  if (tag != mailbox_tag(seq, bits)) return false;
  *bits_out = bits;
  return true;
}

struct Xform {
  float m[12];
};

// 64-bit correspondence key: (float bits of d2) << 32 | target index. For d2 >= 0 the unsigned order of
// the key is (d2, index) lexicographic, so a u64 min merges partial searches and breaks ties on the lowest index.
static constexpr int kGridStatInts = 6;
static constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
static constexpr unsigned int kNoPrev = 0xFFFFFFFFu;  // previous-neighbour position: none (launch_nn_grid_search)

static constexpr int kReduceTerms = 17;   // n, Sp(3), Sq(3), Sqp(9), Sd2
static constexpr int kMaxReduceBlocks = 1024;

struct NnPlan {
  int variant;        // 0 LDS-tiled, 1 scalar-load, 2 LDS + packed f32; +10 = 8 points per lane (tuning knob)
  int splits;         // target splits (grid.y)
  int tgt_per_split;  // multiple of the LDS tile
  int grid_x;
};

NnPlan plan_nn_brute(int n_s, int n_t, int variant, int num_cus);

// keys[i] = min over target of (d2, j) for p = T*src[i]. When plan.splits > 1 keys must be pre-filled with kEmptyKey.
hipError_t launch_nn_brute(const float4* src, int n_s, const float4* tgt, int n_t, const Xform& T, const NnPlan& plan,
                           unsigned long long* keys, hipStream_t stream);

// partials: [blocks][17] doubles, sums_out: 17 doubles (device). Deterministic (fixed order) two-stage reduction.
// The same keys from the matrix cores (icp_brute_mfma.hip).  src_sorted: the source in cell order, ORIGINAL index in .w (a
// grid's sorted copy); keys are written at the original indices and must be pre-filled with kEmptyKey.
// seed (nullable): n_s keys of an earlier sweep of the same source over the SAME target array -- each source's old neighbour
// bounds its search from the first tile on.
hipError_t launch_nn_brute_mfma(const float4* src_sorted, int n_q, const float4* tgt, int n_t, const Xform& T, int num_cus,
                                unsigned long long* keys, const unsigned long long* seed, hipStream_t stream);

// The same keys again with the lower bound on the bf16 matrix path (icp_brute_bf16.hip: split operands, one K = 16 MFMA per
// 32 x 32 pairs, vector work overlapped).  src_morton: launch_morton_order's output.  check (nullable): 2 x u64 zeroed by the
// caller -- the test mode's counters (pairs whose bound was too high; worst excess as float bits in the low half of [1]).
struct GridDesc;
size_t morton_order_work_ints(int n);
hipError_t launch_morton_order(const float4* sorted, int n, const GridDesc& g, int* work, float4* out, hipStream_t stream);
hipError_t launch_nn_brute_bf16(const float4* src_morton, int n_q, const float4* tgt, int n_t, const Xform& T, int num_cus,
                                unsigned long long* keys, const unsigned long long* seed, unsigned long long* check,
                                hipStream_t stream);

// The grid search on the matrix cores (icp_tile.hip): keys[orig source index] = exact NN within sqrt(thr) (else left empty:
// pre-fill with kEmptyKey), found by offering every workgroup of 256 Morton-ordered sources the target points of the grid cells
// their search balls touch, through the bf16 lower-bound filter.  seed (nullable): the previous sweep's keys.
hipError_t launch_nn_tile_search(const float4* src_morton, int n_q, const Xform& T, const float4* sorted, const int* cell_start,
                                 const GridDesc& g, const float4* tgt, int n_t, float thr, const unsigned long long* seed,
                                 float4* prev, bool use_prev, unsigned long long* keys, unsigned long long* stats,
                                 hipStream_t stream);

hipError_t launch_reduce(const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, const Xform& T,
                         float d2_threshold, double* partials, double* sums_out, unsigned long long* flags,
                         unsigned long long seq, hipStream_t stream);

// flags (nullable; 34 64-bit words, host-mapped, 16-byte aligned): term k's workgroup stores the pair {sum bits, seq} at
// words 2k, 2k + 1 in ONE 16-byte write-through store, so the host can poll instead of synchronising the stream and never
// sees a sum without its number; sums_out is then left alone.  Without flags the sums go to sums_out (device memory).
// term_major: partials are laid out [term][block] (what launch_nn_grid_search writes) instead of [block][term].
hipError_t launch_mailbox_selftest(unsigned long long* pair_dev, int rounds, hipStream_t stream);
// n <= 32 ints of device memory as result pairs {value, tag(number)} into mapped host memory (the host polls them)
hipError_t launch_post_ints(const int* d_src, int n, unsigned long long* pairs_dev, unsigned long long number, hipStream_t stream);
hipError_t launch_reduce_final(const double* partials, int n_blocks, bool term_major, double* sums_out,
                               unsigned long long* flags, unsigned long long seq, hipStream_t stream);

hipError_t launch_transform(const float4* src, int n_s, const Xform& T, float4* out, hipStream_t stream);

hipError_t launch_fill_keys(unsigned long long* keys, int n, hipStream_t stream);

// Content fingerprint of a cloud (icpgpu_set_target's recognition of the previous source): the SUM over the points of a
// 64-bit mix of (bits of the point, its index), plus a mix of n -- an order-independent sum, so the host (a loop) and the
// device (any grid) compute the same number.  Two different clouds collide with probability ~2^-64.
ICPGPU_HD inline unsigned long long fp_mix(unsigned long long x) {  // splitmix64's finaliser
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}
ICPGPU_HD inline unsigned long long fp_point(unsigned long long w0, unsigned long long w1, unsigned long long i) {
  return fp_mix(w0 + 0x9e3779b97f4a7c15ull * (2ull * i + 1ull)) + fp_mix(w1 ^ (0xd6e8feb86659fd93ull * (2ull * i + 2ull)));
}
ICPGPU_HD inline unsigned long long fp_finish(unsigned long long sum, unsigned long long n) { return sum + fp_mix(n ^ 0xa5a5a5a5a5a5a5a5ull); }
// *d_acc (zeroed by the launcher) receives the sum of fp_point over pts[0..n)
hipError_t launch_fingerprint(const float4* pts, int n, unsigned long long* d_acc, hipStream_t stream);

// pts[0 .. min(d_counts[0] + d_counts[1], cap)) into host-mapped memory with 16-byte stores, *d_acc (ZERO before: the voxel
// filter's last kernel clears it) += the sum of fp_point over them (the voxel filter's result on its way to the caller: icpgpu_voxel_grid_view); n_most sizes the launch
hipError_t launch_publish_cloud(const float4* pts, const int* d_counts, int n_most, int cap, float4* host_out, unsigned long long* d_acc,
                                hipStream_t stream);

// keys -> (idx, d2) arrays for the kernel-level C-ABI entry point.
hipError_t launch_unpack_keys(const unsigned long long* keys, int n, int32_t* idx, float* d2, hipStream_t stream);

// ---- uniform-grid accelerated exact NN (icp_grid.hip) -------------------------------------------------------
// Target points are counting-sorted by cell (x fastest) into `sorted` (xyz + original index bits in w); a query
// searches the 3x3x3 block of cells around it, then cubes of growing Chebyshev radius, and stops as soon as the
// best squared distance is provably smaller than anything outside the searched cube.
struct GridDesc {
  float ox, oy, oz;  // origin (min corner)
  float h, inv_h;    // cell size
  int nx, ny, nz;
  int sy, sz;        // cell index = cz * sz + cy * sy + cx: x is always the fastest axis (a cell row = one x run); the
                     // thinner of y / z comes next, so that the layers of a flat scene stay close together in memory
  int r_max;         // last cube radius searched: r_max * h * kGridSafety >= cutoff distance
};
static constexpr float kGridSafety = 0.984375f;  // 63/64: covers the rounding of the float binning
static constexpr int kScanItems = 4096;           // elements per block of the cell-count scan

// encoded min/max (6 ints, see decode_bbox) of the finite points of a cloud
hipError_t launch_bbox(const float4* pts, int n, int* d_minmax6, hipStream_t stream, bool init = true);  // init = false: d_minmax6 holds the initial values already
void decode_bbox(const int enc[6], float lo[3], float hi[3]);  // host

// Produces cell_start (exclusive scan of the per-cell counts, ncells+1 entries, in place in counts_then_start),
// sorted[n_valid], d_stats2 = {n_valid, max cell population}.
// Two halves so that the host can look at the occupancy statistics (and pick another cell size) before paying for the
// scan and the scatter.  d_stats (kGridStatInts ints): [0] binned points (valid after finish), [1] largest cell
// population, [2..3] u64 sum of squared populations, [4] binned points (valid after count).
hipError_t launch_grid_count(const float4* pts, int n, const GridDesc& g, int* cell_of_point, int* rank_in_cell,
                             int* counts_then_start, int* block_sums, int* d_stats, hipStream_t stream);
// orig_index (nullable): sorted[].w carries orig_index[i] instead of i (see icpgpu_map_nn_target)
hipError_t launch_grid_finish(const float4* pts, int n, const GridDesc& g, const int* cell_of_point,
                              const int* rank_in_cell, int* counts_then_start, int* block_sums, int* d_stats,
                              const int* orig_index, float4* sorted, hipStream_t stream);

// Exact NN of T*src[i] among the grid's points, guaranteed whenever the NN lies within the cutoff the grid was built
// for; otherwise the point is reported unmatched (empty key).  nn_quad_kernel / nn_wave_kernel (see icp_grid.hip).
//   keys      : optional (nullptr to skip) 8-byte keys with ORIGINAL target indices
//   partials  : optional fused a3+a4 reduction: 17 x grid_search_blocks(n_s) doubles, TERM-major (partials[k * blocks + b])
//   unmatched : optional compaction of unmatched source indices (count at unmatched_count[0], pre-zeroed)
// flags: kGridSrcInCellOrder (XCD-contiguous workgroup mapping), kGridPackShortRows (sparse targets: four short rows per
// step in the cube search), kGridOver4GiB (2^28 or more binned points: nn_quad_kernel's 32-bit byte offsets do not reach,
// nn_wave_kernel takes over)
static constexpr int kGridSrcInCellOrder = 1, kGridPackShortRows = 2, kGridOver4GiB = 4;
// ---- lock-step sweeps over several independent pairs (icpgpu_align_batch) -------------------------------------------
constexpr int kBatchMax = 16;
struct BatchPair {  // what stays the same for a pair from sweep to sweep (a table in device memory)
  const float4* src;
  const float4* sorted;
  const int* cell_start;
  double* partials;
  unsigned int* prev_nn;      // the pair's previous-neighbour positions (below)
  unsigned long long* flags;  // the pair's result mailbox (device alias of pinned host memory)
  unsigned long long* keys;   // ungated (getFitnessScore) sweep: the pair's key array, its list of unmatched points + counter
  int* unmatched;
  int* unmatched_count;
  int r_max_open;             // ... and how far its cubes may grow (nn_keys_grid)
  GridDesc g;
  float accept_thr;
  int n_s, qpw, xcd_map, blocks;
};
struct BatchStep {  // what changes: passed by value with every launch (kernel arguments)
  Xform T[kBatchMax];
  unsigned long long seq[kBatchMax];
  unsigned char slot[kBatchMax];  // blockIdx.y -> row of the table
  unsigned int use_prev_mask;     // bit y: the pair's previous-neighbour buffer holds its last sweep
};
bool grid_search_batchable(int n_s, int flags);
int grid_search_qpw(int n_s);
hipError_t launch_nn_grid_search_batch(const BatchPair* d_pairs, const BatchStep& step, int n_active, int max_blocks, bool pack,
                                       bool open_range, hipStream_t stream);
hipError_t launch_reduce_final_batch(const BatchPair* d_pairs, const BatchStep& step, int n_active, hipStream_t stream);

hipError_t launch_nn_grid_search(const float4* src, int n_s, int flags, const Xform& T, const float4* sorted,
                                 const int* cell_start, const GridDesc& g, float accept_thr, unsigned long long* keys,
                                 double* partials, int* unmatched, int* unmatched_count, hipStream_t stream,
                                 unsigned int* prev_nn = nullptr, bool use_prev = false);
int grid_search_blocks(int n_s);
// Counting runs only (process-wide, not thread-safe): every nn_quad_kernel launch adds the number of target points it
// evaluates to *device_counter; nullptr switches the counting off again.
void grid_count_candidates(unsigned long long* device_counter);
// prev_nn (optional, n_s x 4 bytes): where in `sorted` (byte offset; kNoPrev: nowhere) the neighbour each point found sits,
// written by every sweep of nn_quad_kernel and, with
// use_prev, read back by the next one as an upper bound that prunes its search (valid for ANY transform, but only against
// the same target points and the same src array).  True when launch_nn_grid_search would use it for this size.
bool grid_search_keeps_prev(int n_s, int flags);

// brute force for a list of source indices (fallback for points the grid could not match); keys pre-filled empty.
// the same for at most kFewQueries listed queries, whose number may live in device memory (count_ptr, else n_list): the
// kernel then does nothing for 0 or more than kFewQueries of them and reports the number in *count_out (device-visible)
static constexpr int kFewQueries = 64;
hipError_t launch_nn_brute_few(const float4* src, const int* list, const int* count_ptr, int n_list, const float4* tgt, int n_t,
                               const Xform& T, unsigned long long* keys, int* count_out, hipStream_t stream);
hipError_t launch_nn_brute_list(const float4* src, const int* list, int n_list, const float4* tgt, int n_t,
                                const Xform& T, int num_cus, unsigned long long* keys, hipStream_t stream);

// ---- GICP mode (icp_gicp.hip), SURVEY.md 8(f1) ----------------------------------------------------------------
static constexpr int kGicpK = 20;              // PCL k_correspondences_
static constexpr double kGicpEpsilon = 1e-3;   // PCL gicp_epsilon_
struct Rot3d {
  double m[9];  // row-major
};
static constexpr int kGicpCovFarMost = 1 << 16;  // clouds up to this size: the far-field kernel (a workgroup over the whole cloud per point)
// the covariances of a cloud that has no grid (n <= kGicpCovFarMost): every point through the far-field kernel; list as below
hipError_t launch_gicp_covariances_brute(const float4* cloud, int n, double* cov6, hipStream_t stream, int* list, bool list_counters_zero);
// cov6[i] = upper triangle (xx, xy, xz, yy, yz, zz) of the regularised covariance of point i's 20 nearest neighbours
hipError_t launch_gicp_covariances(const float4* cloud, int n, const float4* sorted, const int* cell_start,
                                   const GridDesc& g, double* cov6, hipStream_t stream, int* list = nullptr, bool list_counters_zero = false);
// (list: 2 n + 2 ints of scratch -- the points the selecting kernel hands to the far-field and the streaming kernel; icp_gicp.hip.
//  list_counters_zero: list[0] and list[1] are zero -- true from the second cloud on the same buffer: the last kernel leaves them so)
// maha6[i] = upper triangle of (C_t[j] + R C_s[i] R^T)^-1 for every source point whose key passes d2 < thr
hipError_t launch_gicp_mahalanobis(int n_s, const unsigned long long* keys, float thr, const Rot3d& R, const double* cov_s,
                                   const double* cov_t, double* maha6, hipStream_t stream);
// One BFGS function/gradient evaluation: at most kGicpDirectBlocks workgroups, each stores its partial sums straight into
// host-mapped memory (host_partials[block * kGicpPartialStride + ...]: [0] = m, [1..13] = high parts of sum r^T M r, sum M r
// (3), sum (base p)(M r)^T (9), [14] = sum d2, [16..28] = the 13 low parts -- the sums are double-double, see icp_gicp.hip)
// followed by host_flags[block] = seq; the host adds the partials in block order.  blocks = gicp_direct_blocks(n_s, share).
// Since round 2 every value travels WITH its sequence number: entry e of a block is the 16-byte pair {value, tag} at doubles
// [2e, 2e + 1], written by ONE 16-byte store -- a pair is never seen half-written, so no flag and no system-scope release
// (a write-back + a wait for the earlier stores' acknowledgements, 1-1.5 us) stands between the sums and the host; the host
// takes a block's entry when its tag equals the evaluation's number.
// Since round 5 a workgroup's answer is four ANSWER LINES of 64 bytes instead of 28 pairs: line L = doubles [8 L, 8 L + 8) of the
// block's stride, seven values (numbers 7 L .. 7 L + 6 of: m, the 13 high parts, sum d2, the 13 low parts) and, in its last word,
// the tag (evaluation number << 24) | XOR of the 24-bit folds of the seven values' bits -- written by ONE store instruction,
// four 64-byte requests instead of seven; the host takes a block's lines when all four tags carry the evaluation's number and
// their checksums (gicp_line_read).
static constexpr int kGicpPartialStride = 64;
static constexpr int kGicpLines = 4;
inline unsigned long long gicp_line_fold(unsigned long long bits) { return (bits ^ (bits >> 24) ^ (bits >> 48)) & 0xFFFFFFull; }
// host side: value number s (0 = m, 1..13 high parts, 14 = sum d2, 15..27 low parts) of a block whose lines have been validated
inline double gicp_line_value(const volatile double* block, int s) { return block[8 * (s / 7) + (s % 7)]; }
static constexpr int kGicpDirectBlocks = 256;  // capacity of the mailbox; gicp_direct_blocks() may use fewer
int gicp_direct_blocks(int n_s, int most = kGicpDirectBlocks);
hipError_t launch_gicp_cost_direct(int blocks, const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                                   const Xform& T, const Xform& base, const double* maha6, double* host_partials,
                                   unsigned long long* host_flags, unsigned long long seq, hipStream_t stream);

// Resident variant for a whole BFGS run (icp_gicp.hip): the kernel waits for commands in a line of fine-grained device
// memory the host writes through the BAR (cmd: 12 floats of T, then the sequence number), evaluates, answers through
// host_partials / host_flags like the direct kernel (flag = seq_hi << 32 | sequence number) and leaves on sequence number
// kGicpServerExit (acknowledged by host_flags[0] = ~0) or after 50 ms without a command.
// The device solver (icp_gicp.hip: gicp_solve_kernel): the whole BFGS run of an outer iteration inside one resident kernel.
// slots: gicp_solve_slot_bytes(blocks) of FINE-GRAINED device memory; host_out: gicp_solve_out_granules() granules (16 B each) of
// the host mailbox -- granule 0 the status (gicp::Status), 1..6 the state, 7 m, 8 sum d2, 9 f at the start, 10 evaluations --
// each carrying number seq0 and a checksum (gicp_granule_read).  Evaluation e of the run uses number seq0 + e.
size_t gicp_solve_slot_bytes(int blocks);
int gicp_solve_out_granules();
hipError_t launch_gicp_solve(int blocks, const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                             const Xform& base, const float guess[16], const double* maha6, const double x0[6],
                             unsigned long long* slots, unsigned long long* host_out, unsigned long long seq0, int max_inner,
                             double gradient_tol, hipStream_t stream, unsigned long long* local_slots = nullptr,
                             unsigned long long* owner = nullptr, int xcc_want = 0);
// one-XCD variant (local_slots / owner given, <= gicp_solve_local_blocks() workgroups, correspondences resident): ordinary device
// memory (gicp_solve_slot_bytes) for the granules, kGicpDirectBlocks words for the worker claims
int gicp_solve_local_blocks();
// Several runs' device solvers in one launch (gicp_solve_batch_kernel): every item a RESIDENT run (blocks x 1024 >= n_s) with its
// own slots / host_out / numbers, as launch_gicp_solve takes them (seq0 may carry kMailboxReleaseBit); any placement.
constexpr int kGicpSolveBatchMax = 8;
struct GicpSolveItem {
  const float4* src;
  int n_s;
  const float4* tgt;
  const unsigned long long* keys;
  float thr;
  Xform base;
  float guess[16];
  const double* maha6;
  double x0[6];
  unsigned long long* slots;
  unsigned long long* host_out;
  unsigned long long seq0;
  int blocks;
};
hipError_t launch_gicp_solve_batch(const GicpSolveItem* items, int n, int max_inner, double gradient_tol, hipStream_t stream);
bool gicp_granule_read(const volatile unsigned long long* g, unsigned long long seq, double* value);
// The quadratic form of an outer iteration (icp_gicp_quadratic.h, icp_gicp.hip: gicp_quadratic_kernel): kGicpQuadSums
// double-double sums over the correspondences whose key passes d2 < thr, Mahalanobis matrices computed on the way (R: the rotation
// launch_gicp_mahalanobis takes).  partials: gicp_quadratic_blocks(n_s) x kGicpQuadSums x 2 doubles of device memory, done: a zeroed
// device word; host_out: 2 x kGicpQuadSums result pairs (16 B each) numbered seq -- pair 2 n the high part of sum n, 2 n + 1 its low
// part (gicp_granule_read).
static constexpr int kGicpQuadSums = 75;
static constexpr int kGicpQuadBlocks = 256;
int gicp_quadratic_blocks(int n_s);
hipError_t launch_gicp_quadratic(const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                                 const Rot3d& R, const double* cov_s, const double* cov_t, double* partials, unsigned int* done,
                                 unsigned long long* host_out, unsigned long long seq, hipStream_t stream);
static constexpr unsigned int kGicpServerExit = 0xFFFFFFFFu;
hipError_t launch_gicp_server(int blocks, const float4* src, int n_s, const float4* tgt, const unsigned long long* keys, float thr,
                              const Xform& base, const double* maha6, double* host_partials, unsigned long long* host_flags,
                              unsigned int* cmd, unsigned int first_seq, unsigned int seq_hi, hipStream_t stream);

// ---- the mapper's one-point-per-voxel map (icp_map.hip), SURVEY.md 8(f4) --------------------------------------------
struct MapDesc {
  double ox, oy, oz;  // lattice origin = first inserted point - resolution (PCL OctreePointCloud: first box p +- res/2, widened to 2 voxels by getKeyBitSize)
  double res;         // voxel size (octree_resolution_, 0.5 m)
};
// PCL's octree geometry for the faithful approxNearestSearch mode (icp_map.hip): the bounding box as
// OctreePointCloud::adoptBoundingBoxToPoint has grown it, in double; depth = levels below the root (side = 2^depth voxels)
struct ApproxBox {
  double min[3], max[3];
  double res;
  int depth;
};
// PCL assigns a point's leaf key ONCE, from the bounding box of the moment it is added (genOctreeKeyforPoint), and re-roots the
// tree when the box doubles -- a stored key then moves by whole voxels with the minimum; it is never recomputed from the new
// minimum (which could round the other way for a point on a voxel border: the FIRST point sits on one by construction, the
// first box being centred on it).  The history of the box: version v is in force from map index first[v] on; a point added
// under it has key trunc((p - min[v]) / res) + (shift_now - shift[v]), shifts in voxels.
constexpr int kApproxMaxVersions = 24;
struct ApproxHistory {
  int n;
  int first[kApproxMaxVersions];
  double min[kApproxMaxVersions][3];
  long long shift[kApproxMaxVersions][3];
};
hipError_t launch_approx_first_outside(const float4* pts, int n, const ApproxBox& b, int* d_first, hipStream_t stream);
hipError_t launch_approx_fill(unsigned long long* keys, int* vals, unsigned int cap, hipStream_t stream);
hipError_t launch_approx_insert(const float4* pts, int lo, int hi, const ApproxBox& b, const ApproxHistory& h, unsigned long long* keys,
                                int* vals, unsigned int cap, hipStream_t stream);
hipError_t launch_approx_descend(const float4* queries, int n, const Xform& T, const ApproxBox& b, const unsigned long long* keys,
                                 const int* vals, unsigned int cap, unsigned long long* out, hipStream_t stream);
int approx_max_depth();
// generic primitives (icp_scan.hip): exclusive prefix sum of int32 (scratch: exclusive_scan_scratch_ints(n) ints) and a stable
// LSD radix sort of (key, value) int32 pairs from the first halves of keys / vals (2 n ints each) into the second halves
size_t exclusive_scan_scratch_ints(int n);
hipError_t launch_exclusive_scan(const int* in, int* out, int n, int* scratch, hipStream_t stream);
size_t radix_sort_scratch_ints(int n);
hipError_t launch_radix_sort_pairs(int* keys, int* vals, int n, unsigned int end_bit, int* scratch, hipStream_t stream);
size_t map_scan_temp_bytes(int n);
// hash set: keys (packed voxel coordinates, all-ones = empty), vals (map index, -1 = claimed this call), first (bids)
hipError_t launch_map_fill(unsigned long long* keys, int* vals, int* first, unsigned int cap, hipStream_t stream);
hipError_t launch_map_rehash(const float4* map_pts, int n_map, const MapDesc& m, unsigned long long* keys, int* vals,
                             unsigned int cap, hipStream_t stream);
// addPointsToMap(): p = T * in[i]; the first point (lowest i) of every unoccupied voxel is appended to map_pts at
// base + (its rank among the appended), input order preserved; *d_n_added = number appended.  cap is a power of two.
hipError_t launch_map_insert(const float4* in, int n, const Xform& T, const MapDesc& m, unsigned long long* keys, int* vals,
                             int* first, unsigned int cap, float4* moved, int* slot_of, int* flags, int* rank, void* temp,
                             size_t temp_bytes, int base, float4* map_pts, int* d_n_added, hipStream_t stream);
// nn cloud: out = T_out * map[index(keys[i])] for every non-empty key, order preserved; *d_n_out = points written
hipError_t launch_map_nn_gather(const unsigned long long* keys, int n, const float4* map_pts, const Xform& T_out, int* flags,
                                int* rank, void* temp, size_t temp_bytes, float4* out, int* d_n_out, hipStream_t stream);
// The nn cloud repeats every map point once per scan point that chose it.  uniq / uniq_index receive its DISTINCT points
// (first occurrences, in nn-cloud order) and their positions in the nn cloud; first_user is n_map ints of scratch,
// uflags / urank n ints each.  flags / rank are the arrays launch_map_nn_gather left behind.
hipError_t launch_map_nn_unique(const unsigned long long* keys, const int* flags, const int* rank, int n, const float4* nn_cloud,
                                int n_map, int* first_user, int* uflags, int* urank, void* temp, size_t temp_bytes,
                                float4* uniq, int* uniq_index, int* d_n_uniq, hipStream_t stream);

// ---- voxel-grid down-sampling (icp_voxel.hip), SURVEY.md 8(f2) -------------------------------------------------
size_t voxel_temp_bytes(int n);
// keys/vals: 2*n ints each, flags/slots: n ints each, d_n_out: 2 ints whose sum is the number of cells written to `out`.
hipError_t launch_voxel_grid(const float4* pts, int n, float inv_leaf, const int minb[3], const int divb[3], int* keys,
                             int* vals, int* flags, int* slots, void* temp, size_t temp_bytes, float4* out, int* d_n_out,
                             hipStream_t stream);

// The same filter without the library sort (one distribution pass + a sort in LDS; icp_voxel.hip): bins =
// voxel_direct_scratch_ints(n) ints, all zero before the first call; published = voxel_direct_groups(n) 64-bit words used for
// nothing else, zero before the first call; keys, relpos n ints each; comp n 64-bit words;
// *status != 0 when a voxel bucket exceeded the LDS capacity -- nothing usable was written, run launch_voxel_grid instead.
size_t voxel_direct_scratch_ints(int n);
int voxel_direct_groups(int n);
hipError_t launch_voxel_grid_direct(const float4* pts, int n, float inv_leaf, const int minb[3], const int divb[3], int* bins,
                                    unsigned long long* published, int* keys, int* relpos, unsigned long long* comp, float4* out,
                                    int* d_n_out, int* status, hipStream_t stream, unsigned long long* clear_word = nullptr,
                                    int* d_bbox6 = nullptr, int* d_plan = nullptr);
// (d_plan, 14 ints, with d_bbox6 = launch_bbox's result queued in front: the device derives minb / divb / the buckets itself and the
// host need not have seen the box; plan[7] = 0 the filter ran, 1 no finite point, 2 PCL's pass-through, 3 the sort path's cloud;
// plan[8..13] = the box, d_bbox6 itself is left as bbox_init_kernel leaves it: the next launch_bbox on it may skip its init)
// (clear_word, optional: a 64-bit word of device memory the last kernel sets to zero -- launch_publish_cloud's accumulator)

}  // namespace icpgpu
