/*
 * icpgpu.h -- C-ABI of libicpgpu.so: MI355X (gfx950) ICP scan-matching core.
 *
 * Drop-in boundary for the one data-parallel hot path of YoshuaNava/icpslam: the per-scan
 * registration the reference delegates to a PCL `Registration` object at
 *   /root/reference/src/icpslam/icp_odometer.cpp:188-201   (IcpOdometer::laserCloudCallback)
 *   /root/reference/src/icpslam/octree_mapper.cpp:104-117  (OctreeMapper::estimateTransformICP)
 * The reference has no FFI/plugin registry; the interface it binds is the 10-call PCL protocol
 *   ctor -> setMaximumIterations -> setTransformationEpsilon -> setMaxCorrespondenceDistance ->
 *   setRANSACIterations(0) -> setInputSource -> setInputTarget -> align(out) ->
 *   getFinalTransformation -> hasConverged -> getFitnessScore.
 * Each entry point below names the protocol call (reference file:line) it replaces.  A header-only
 * C++ shim with exactly those method names lives in include/icpgpu_registration.hpp; the
 * reference-side edit is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; every function returns an int status
 *     (0 = ICPGPU_OK, < 0 = error) and never throws or aborts.  Non-convergence is NOT an error:
 *     it is reported as result.converged == 0, like PCL's hasConverged().
 *   - clouds are `pcl::PointXYZ` arrays: 16-byte stride float {x, y, z, pad}; pad is ignored on
 *     input and written as 1.0f on output.
 *   - transforms are float[16] column-major 4x4 (memory layout of Eigen::Matrix4f), mapping
 *     source -> target, i.e. the T the reference chains at icp_odometer.cpp:112-113.
 *   - one context = one device + one HIP stream + scratch; distinct contexts may be used from
 *     different threads concurrently (the odometer callback thread and the mapper main-loop
 *     thread, /root/reference/src/icpslam_node.cpp:9); a single context is not re-entrant.
 *   - there is no CPU fallback: without a usable gfx950 device icpgpu_create() fails.
 */
#ifndef ICPGPU_H
#define ICPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPGPU_VERSION_MAJOR 1
#define ICPGPU_VERSION_MINOR 1
#define ICPGPU_HEADER_VERSION (ICPGPU_VERSION_MAJOR * 1000 + ICPGPU_VERSION_MINOR)
/* ABI rule (1.0).  icpgpu_params, icpgpu_result and icpgpu_profile only ever GROW AT THE END, and the library never assumes the
 * caller's structs are as long as its own: the caller's sizeof of the three travels with icpgpu_create (the macro below hands them
 * to icpgpu_create_abi) and every entry point that reads or writes one of them through that context copies min(caller's, library's)
 * bytes -- fields the caller does not know are not written, fields the library does not know read as zero.  The two entry points
 * without a context take the sizes themselves (icpgpu_default_params_sz, icpgpu_align_batch_multi_sz; macros below).
 * icpgpu_create_abi refuses a header of another MAJOR version (ICPGPU_ERR_UNSUPPORTED).  Until 0.4 the structs grew in place with
 * nothing but a comment to protect an older caller; the unsized symbols of those versions (icpgpu_create, icpgpu_default_params,
 * icpgpu_align_batch_multi) are NOT exported any more, so a binary built against a 0.x header fails at load time instead of
 * overrunning its structs.  History: 1.1 icpgpu_align_view, icpgpu_voxel_grid_view (result clouds as views of the pinned staging
 * buffer), icpgpu_profile.voxel_views_direct; 1.0 icpgpu_result.gicp_solver, icpgpu_calibrate, sized entry points; 0.4 icpgpu_params.gicp_inner,
 * icpgpu_profile.gicp_quadratic_solves; 0.3 icpgpu_profile (sources_adopted, gicp_host_solves, gicp_solver_choice). */

/* ---- environment ---------------------------------------------------------------------------------
 * Production switches, read by every build of libicpgpu.so (none of them changes a result, except ICPGPU_GICP_INNER):
 *   ICPGPU_WAIT_TIMEOUT_MS        deadline of every host wait for the device (mailbox, gather); default 30000
 *   ICPGPU_BATCH_THREADS          host threads of icpgpu_align_batch (default: chosen from the CPUs this process may use)
 *   ICPGPU_BATCH_DEPTH            pairs per lock-step group (point-to-point; default 8) / alignments a thread keeps in flight
 *   ICPGPU_BATCH_GROUPS           lock-step groups in flight on the GPU, over all host threads together (default 8)
 *   ICPGPU_RECOGNISE=0            icpgpu_set_target / icpgpu_set_source always upload (no content recognition)
 *   ICPGPU_GICP_SERVER=0          every GICP cost evaluation is its own kernel launch (no resident evaluation server)
 *   ICPGPU_GICP_DEVICE=0|1|auto   GICP's inner BFGS on the host over the evaluation server (0), or in the resident device solver (1);
 *                                 auto (the default) = the host loop for single alignments, the device solver for the runs of a
 *                                 batch -- a fixed rule since 1.0 (until 0.4 a context timed both over its first alignments);
 *                                 icpgpu_calibrate measures on request.  Same bits either way; the solver an alignment ran on is
 *                                 in icpgpu_result.gicp_solver, the context's setting in icpgpu_profile.gicp_solver_choice
 *   ICPGPU_GICP_INNER=exact|quadratic  overrides icpgpu_params.gicp_inner (see icpgpu_gicp_inner below: QUADRATIC moves a GICP
 *                                 result within the stated tolerance, not bit for bit)
 *   ICPGPU_MAILBOX=pairs|release  how results reach the host (default: a self-test at context creation picks it)
 *   ICPGPU_DEBUG=1                diagnostics on stderr
 *   LOCAL_WORLD_SIZE              (torch.distributed.run) processes sharing this host's CPUs
 * Development switches (kernel variants, tuning constants, test modes) exist only in libicpgpu_dev.so: icpslam_amd/csrc/icp_env.h. */

typedef struct icpgpu_ctx icpgpu_ctx; /* opaque */

typedef enum {
  ICPGPU_OK = 0,
  ICPGPU_ERR_INVALID_ARG = -1,
  ICPGPU_ERR_NO_DEVICE = -2,   /* no HIP device / not gfx950 / runtime missing */
  ICPGPU_ERR_HIP = -3,         /* a HIP call failed; see icpgpu_last_error() */
  ICPGPU_ERR_OOM = -4,
  ICPGPU_ERR_NO_INPUT = -5,    /* align/fitness before set_source/set_target */
  ICPGPU_ERR_UNSUPPORTED = -6
} icpgpu_status;

/* Solver. The reference instantiates pcl::GeneralizedIterativeClosestPoint (icp_odometer.cpp:188,
 * octree_mapper.cpp:104); BASELINE.json's north_star specifies point-to-point ICP
 * (pcl::IterativeClosestPoint semantics), which is the primary mode. */
typedef enum { ICPGPU_P2P_SVD = 0, ICPGPU_GICP = 1 } icpgpu_method;

/* GICP's inner minimisation (PCL: estimateRigidTransformationBFGS, ~35 cost evaluations per outer iteration).
 *   EXACT      every evaluation is a pass over the correspondences with PCL's arithmetic (points transformed in float32); the
 *              registration is bit-identical to the CPU restatement the tests compare with.  The default.
 *   QUADRATIC  the correspondences are reduced ONCE per outer iteration to the 73 coefficients of the quadratic form the cost is
 *              when the transformed points are taken as real numbers; BFGS -- the same solver -- then runs on the host without a
 *              device round trip (2-3x the scans/s of the reference's pipeline).  The result differs from EXACT's by what PCL's
 *              float32 rounding of the transformed points contributes: within 1e-4 / 1e-3 m on most pairs, about as far from
 *              PCL's evaluation as that evaluation moves when its own sums are re-ordered (profiles/r05_gicp_quadratic.txt).
 * ICPGPU_GICP_INNER=exact|quadratic in the environment overrides the parameter (an unchanged binary can opt in). */
typedef enum { ICPGPU_GICP_INNER_EXACT = 0, ICPGPU_GICP_INNER_QUADRATIC = 1 } icpgpu_gicp_inner;

/* Correspondence search strategy; every mode returns the exact nearest neighbour. */
typedef enum {
  ICPGPU_NN_AUTO = 0,
  ICPGPU_NN_BRUTE = 1, /* LDS-tiled brute force (north_star) */
  ICPGPU_NN_GRID = 2   /* uniform-grid accelerated exact search */
} icpgpu_nn_mode;

/* pcl::registration::DefaultConvergenceCriteria::ConvergenceState */
typedef enum {
  ICPGPU_NOT_CONVERGED = 0,
  ICPGPU_CONV_ITERATIONS = 1,
  ICPGPU_CONV_TRANSFORM = 2,
  ICPGPU_CONV_ABS_MSE = 3,
  ICPGPU_CONV_REL_MSE = 4,
  ICPGPU_CONV_NO_CORRESPONDENCES = 5
} icpgpu_convergence_state;

typedef struct {
  int32_t method;                     /* icpgpu_method */
  int32_t max_iterations;             /* setMaximumIterations: icp_odometer.cpp:189 (10), octree_mapper.cpp:105 (30) */
  double transformation_epsilon;      /* setTransformationEpsilon: icp_odometer.cpp:190 (1e-6) */
  double max_correspondence_distance; /* setMaxCorrespondenceDistance: icp_odometer.cpp:191 (1.0 m) */
  double euclidean_fitness_epsilon;   /* PCL default -DBL_MAX: relative-MSE test off (never set by the reference) */
  int32_t min_correspondences;        /* PCL default 3 (P2P) / 4 (GICP) */
  int32_t force_iterations;           /* != 0: run exactly max_iterations (benchmarking; no PCL equivalent) */
  int32_t nn_mode;                    /* icpgpu_nn_mode */
  int32_t brute_variant;              /* brute-force search: 0 = matrix-core kernel for large clouds (default: lower bound on the bf16
                                       * matrix path), 1 = plain-VALU kernel always, 2 = the bound in f32 MFMAs; identical results */
  int32_t gicp_inner;                 /* icpgpu_gicp_inner (GICP only); 0 = EXACT */
} icpgpu_params;

typedef struct {
  float T[16];                /* getFinalTransformation(): icp_odometer.cpp:199, octree_mapper.cpp:115 */
  int32_t converged;          /* hasConverged(): icp_odometer.cpp:201, octree_mapper.cpp:117 */
  int32_t iterations;
  int32_t convergence_state;  /* icpgpu_convergence_state */
  uint32_t n_correspondences; /* accepted pairs in the last iteration */
  double mse_last;            /* mean squared correspondence distance of the last iteration (m^2) */
  double fitness;             /* getFitnessScore(): icp_odometer.cpp:201; NaN unless requested */
  double t_total_ms;          /* host wall time of the align call */
  double t_device_ms;         /* kernel time of the call on the context's stream: HIP-event time of the TIMED sweeps
                               * (icpgpu_profile_set_sampling), scaled to all sweeps of the call.  An estimate: timings are
                               * read without blocking, and one that was not ready at the end of the call counts towards
                               * the next one; 0 when none of the call's sweeps was a timed one (icpgpu_profile_get always waits and is exact) */
  int32_t gicp_solver;        /* GICP: where the inner BFGS of this alignment's LAST outer iteration ran -- icpgpu_gicp_solver;
                               * 0 for point-to-point alignments and alignments that never reached a minimisation -- 1.0 */
  int32_t reserved0;
} icpgpu_result;
typedef enum {
  ICPGPU_GICP_SOLVER_NONE = 0,
  ICPGPU_GICP_SOLVER_HOST = 1,      /* BFGS on the host, every evaluation a round trip to the resident evaluation server */
  ICPGPU_GICP_SOLVER_DEVICE = 2,    /* the whole BFGS inside gicp_solve_kernel */
  ICPGPU_GICP_SOLVER_QUADRATIC = 3  /* gicp_inner = QUADRATIC: one device pass, BFGS on the host on the quadratic form */
} icpgpu_gicp_solver;

/* Kernel-level accounting since the last icpgpu_profile_reset(); times are HIP-event times on the
 * context's own stream (this is what bench.py's roofline object is computed from). */
typedef struct {
  uint64_t nn_launches;       /* brute-force correspondence-search launches (a2) */
  double nn_ms;               /* summed duration of the TIMED ones (nn_timed of them, see icpgpu_profile_set_sampling) */
  uint64_t nn_pairs;          /* point pairs evaluated by those launches */
  uint64_t nn_bytes;          /* algorithmic bytes: 16*(N_s+N_t) + 8*N_s per launch */
  uint64_t reduce_launches;   /* rejection + covariance reduction launches (a3+a4) */
  double reduce_ms;
  uint64_t reduce_bytes;
  uint64_t transform_launches; /* a6 */
  double transform_ms;
  uint64_t transform_bytes;
  uint64_t iterations;        /* ICP iterations executed */
  uint64_t aligns;            /* align calls */
  uint64_t grid_launches;     /* grid-accelerated correspondence launches (a2, with a3+a4 fused inside ICP iterations) */
  double grid_ms;
  uint64_t grid_bytes;        /* algorithmic bytes: 16*(N_s+N_t) + output per launch */
  uint64_t grid_builds;       /* target grid (re)builds: bbox + counting sort */
  double grid_build_ms;       /* host time inside the builds (two host round trips each; the last kernels overlap the caller) */
  uint64_t grid_fallback_points; /* points finished by the brute-force kernel (no neighbour within the cutoff) */
  uint64_t voxel_launches;    /* voxel-grid filter runs */
  double voxel_ms;            /* key + sort + flag/scan + centroid kernels */
  uint64_t voxel_bytes;       /* algorithmic bytes: 16*N in + 16*N_out */
  uint64_t gicp_cov_launches; /* GICP: per-cloud 20-NN covariance passes */
  double gicp_cov_ms;
  uint64_t gicp_cost_launches; /* GICP: BFGS function/gradient evaluations (one device reduction each; with gicp_inner = QUADRATIC they
                               * run on the host, and gicp_eval_ms then holds the passes' and the minimisations' wall time) */
  uint64_t map_inserts;       /* f4: addPointsToMap batches */
  double map_insert_ms;
  uint64_t map_points_in;     /* points offered to the map */
  uint64_t map_nn_launches;   /* f4: nn-cloud builds */
  double map_nn_ms;
  uint64_t nn_timed;          /* launches behind nn_ms / grid_ms / reduce_ms: kernel timing is sampled, because the three */
  uint64_t grid_timed;        /*   event records around a sweep are barrier packets that cost 6-7 us per iteration */
  uint64_t reduce_timed;
  uint64_t grid_bounded;      /* grid sweeps whose searches were pruned by the neighbours the previous sweep found */
  double gicp_eval_ms;        /* GICP: host wall time of the BFGS cost evaluations, command written -> 13 sums merged (the
                               * evaluations are dependent host <-> device round trips; this is what a registration waits for) */
  uint64_t gicp_eval_corr;    /* correspondences those evaluations reduced over, summed (88 algorithmic bytes each) */
  uint64_t gicp_cov_points;   /* points whose covariances the gicp_cov passes computed (16 B read + 48 B written each, + the 20-NN search) */
  uint64_t targets_recognised; /* icpgpu_set_target calls that found the cloud already in HBM (no upload, no rebuild) */
  uint64_t brute_bound_violations; /* test mode ICPGPU_MFMA_CHECK_BOUND=1 of the bf16 matrix-core search: pairs whose lower bound */
  double brute_bound_worst;        /*   exceeded what their own exact distance allows (must stay 0); worst excess / (P^2 + |v|^2) seen */
  uint64_t gicp_device_solves;     /* GICP outer iterations whose whole inner BFGS ran on the device (gicp_solve_kernel) */
  uint64_t grid_adopted;           /* GICP: targets whose correspondence search took over the grid their covariances were computed
                                    * over instead of building a second one (same keys: the search is exact whatever the cells) */
  uint64_t sources_adopted;        /* icpgpu_set_source calls that found the buffer to be the context's last voxel-filter result,
                                    * still in HBM (no upload, no bounding-box pass) -- version 0.3 */
  uint64_t gicp_host_solves;       /* GICP outer iterations whose inner BFGS ran on the host (over the evaluation server) -- 0.3 */
  uint64_t gicp_solver_choice;     /* the solver single GICP alignments of this context run on: 1 = host loop, 2 = device solver; never 0
                                    * since 1.0 (set at icpgpu_create from ICPGPU_GICP_DEVICE, changed only by icpgpu_calibrate) -- 0.3 */
  uint64_t gicp_quadratic_solves;  /* GICP outer iterations solved on the quadratic form (icpgpu_params.gicp_inner = QUADRATIC) -- 0.4 */
  uint64_t cov_grids_unchecked;    /* GICP: covariance grids built without waiting for their occupancy statistics (a containing box from
                                    * the voxel filter + the last cloud's cell size; the statistics are checked at the alignment's first
                                    * wait) -- 1.0 */
  uint64_t cov_grids_rebuilt;      /* ... of which the check failed: rebuilt the waiting way, the alignment started over -- 1.0 */
  uint64_t voxel_views_direct;     /* icpgpu_voxel_grid_view calls whose points reached the host in front of the cell count (the others
                                    * went through the copy engine: sort path, pass-through, a result beyond the staging buffer) -- 1.1 */
} icpgpu_profile;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* replaces: construction of the stack `icp` object, icp_odometer.cpp:188 / octree_mapper.cpp:104.
 * Unlike the reference (fresh object per scan) a context is meant to be created once and reused. */
int icpgpu_create_abi(icpgpu_ctx** out_ctx, int device_id, int header_version, size_t sizeof_params, size_t sizeof_result,
                      size_t sizeof_profile);
#define icpgpu_create(out_ctx, device_id) \
  icpgpu_create_abi((out_ctx), (device_id), ICPGPU_HEADER_VERSION, sizeof(icpgpu_params), sizeof(icpgpu_result), sizeof(icpgpu_profile))
int icpgpu_destroy(icpgpu_ctx* ctx);
const char* icpgpu_last_error(const icpgpu_ctx* ctx); /* ctx may be NULL: last create() error */
int icpgpu_version(void);                             /* major*1000 + minor */

/* ---- parameters (icp_odometer.cpp:189-192, octree_mapper.cpp:105-108) ----------------------- */
void icpgpu_default_params_sz(icpgpu_params* p, size_t sizeof_params); /* PCL defaults + the reference's odometer constants */
#define icpgpu_default_params(p) icpgpu_default_params_sz((p), sizeof(icpgpu_params))
/* what THIS library's structs measure: {sizeof(icpgpu_params), sizeof(icpgpu_result), sizeof(icpgpu_profile)} (bindings that
 * mirror the structs by hand -- ctypes, JNA -- check themselves against it) */
void icpgpu_struct_sizes(size_t out3[3]);
int icpgpu_set_params(icpgpu_ctx* ctx, const icpgpu_params* p);
int icpgpu_get_params(const icpgpu_ctx* ctx, icpgpu_params* p);

/* ---- inputs --------------------------------------------------------------------------------- */
/* replaces setInputSource (icp_odometer.cpp:193, octree_mapper.cpp:109): copies n points H2D;
 * the caller keeps ownership. n may be 0. */
int icpgpu_set_source(icpgpu_ctx* ctx, const float* xyzw, size_t n);
/* replaces setInputTarget (icp_odometer.cpp:194, octree_mapper.cpp:110). PCL rejects an empty
 * target; here n == 0 is accepted and align() then reports converged = 0 with T = identity. */
int icpgpu_set_target(icpgpu_ctx* ctx, const float* xyzw, size_t n);
/* The reference hands scan k-1's SOURCE cloud back as scan k's TARGET (`*prev_cloud_ = *curr_cloud_`,
 * icp_odometer.cpp:209, then setInputTarget(prev_cloud_) at :194): icpgpu_set_target recognises a buffer
 * whose content (size, then a 64-bit content fingerprint) is the context's current source or target and
 * keeps what it has in HBM -- the cloud, its search grid, its GICP covariances -- instead of uploading
 * and rebuilding (the promote path for the source; nothing at all for the unchanged target of a rejected
 * scan); the source stays set (as a device-side copy) either way.  Call set_target BEFORE set_source when both change
 * (after set_source the previous source is gone and there is nothing to recognise).  icpgpu_fingerprint is that fingerprint of
 * a host buffer (n points of 16 bytes); icpgpu_cloud_sizes reports what a context holds (the C++ shim's
 * context pool picks the context whose source has the new target's size).
 * ASSUMPTION: recognition compares sizes and the 64-bit fingerprint (an additive, non-cryptographic mix of every point's bits
 * and index), not the bytes: two different clouds of equal size collide with probability ~2^-64 per comparison, and an
 * adversarial cloud could be constructed.  ICPGPU_RECOGNISE=0 in the environment makes icpgpu_set_target always upload.
 * icpgpu_set_source recognises in the same way (and under the same switch) the result of the context's last icpgpu_voxel_grid
 * when the caller hands it back: see icpgpu_voxel_grid below. */
unsigned long long icpgpu_fingerprint(const float* xyzw, size_t n);
int icpgpu_cloud_sizes(const icpgpu_ctx* ctx, size_t* n_source, size_t* n_target);
/* same, for clouds already resident in this device's HBM (zero copy; must stay valid and
 * unmodified until replaced; 16-byte aligned). */
int icpgpu_set_source_device(icpgpu_ctx* ctx, const void* d_xyzw, size_t n);
int icpgpu_set_target_device(icpgpu_ctx* ctx, const void* d_xyzw, size_t n);
/* make the current source the next target without a copy (the reference's
 * `*prev_cloud_ = *curr_cloud_`, icp_odometer.cpp:209). */
int icpgpu_promote_source_to_target(icpgpu_ctx* ctx);

/* ---- the hot path --------------------------------------------------------------------------- */
/* replaces align(out) + getFinalTransformation + hasConverged [+ getFitnessScore]
 * (icp_odometer.cpp:198-201, octree_mapper.cpp:114-117).
 *   guess     : float[16] initial transform or NULL (= identity; the reference never passes one)
 *   out_xyzw  : host buffer for the aligned source cloud (n_source * 16 bytes) or NULL to skip the
 *               D2H copy (the reference only publishes it for debugging, icp_odometer.cpp:216-218)
 *   want_fitness : != 0 also evaluates getFitnessScore() (one more NN sweep)                     */
int icpgpu_align(icpgpu_ctx* ctx, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* result);
/* the same, the aligned cloud handed out as a VIEW: *view_xyzw points at *n_out points in the context's pinned staging buffer
 * (written there by the transform kernel itself, in front of the fitness sweep), valid until the context's next call.  For a
 * caller that owns a container to fill -- align(output) resizes `output` (icp_odometer.cpp:196-198): `output.points.assign(view,
 * view + n)` is one pass where resize + copy are two.  1.1 */
int icpgpu_align_view(icpgpu_ctx* ctx, const float* guess, int want_fitness, icpgpu_result* result, const float** view_xyzw,
                      size_t* n_out);

/* replaces getFitnessScore(max_range) (icp_odometer.cpp:201) using the last align's transform. */
int icpgpu_fitness(icpgpu_ctx* ctx, double max_range, double* out_fitness);

/* Many independent scan pairs through one context (BASELINE config 4/5).  Pair k registers
 * src[k] (n_src[k] points) onto tgt[k]; all pointers are host pointers.  Iterations of different
 * pairs are interleaved on the device so the per-iteration host solve of one pair overlaps the
 * kernels of others.  results[k] is filled for every k. */
int icpgpu_align_batch(icpgpu_ctx* ctx, size_t n_pairs, const float* const* src, const size_t* n_src,
                       const float* const* tgt, const size_t* n_tgt, int want_fitness, icpgpu_result* results);

/* The same over the GPUs of one node, from ONE process (the reference is a single C++ process,
 * /root/reference/src/icpslam_node.cpp:3-14): entry r of `devices` gets the contiguous shard r of the pairs (sizes differ by
 * at most one) and its own host thread driving icpgpu_align_batch on a context the library keeps for that entry; no data
 * is exchanged during the solve.  results[k] is filled for every k.  With a communicator the result records -- 23 float64 =
 * 184 B per pair: pair id, iterations, converged, state, n_corr, mse, fitness, T[16] row-major -- are then all-gathered
 * across the devices (each shard padded to the largest) and the checked copy of entry 0 is returned in `records`
 * (n_pairs x 23 doubles):
 *   ICPGPU_COMM_NONE  no gather (records may be NULL)
 *   ICPGPU_COMM_RCCL  ncclCommInitAll over `devices` + one ncclAllGather (librccl is loaded on first use, not linked)
 *   ICPGPU_COMM_HOST  the same buffers, exchanged through host memory (tests on one GPU: a device may be named twice)
 * params NULL keeps the contexts' parameters.  Errors: status code + icpgpu_multi_last_error() (per calling thread). */
enum { ICPGPU_COMM_NONE = 0, ICPGPU_COMM_RCCL = 1, ICPGPU_COMM_HOST = 2 };
int icpgpu_align_batch_multi_sz(const int* devices, int n_devices, const icpgpu_params* params, size_t n_pairs,
                                const float* const* src, const size_t* n_src, const float* const* tgt, const size_t* n_tgt,
                                int want_fitness, icpgpu_result* results, double* records, int communicator, size_t sizeof_params,
                                size_t sizeof_result);
#define icpgpu_align_batch_multi(devices, n_devices, params, n_pairs, src, n_src, tgt, n_tgt, want_fitness, results, records, comm) \
  icpgpu_align_batch_multi_sz((devices), (n_devices), (params), (n_pairs), (src), (n_src), (tgt), (n_tgt), (want_fitness), (results), \
                              (records), (comm), sizeof(icpgpu_params), sizeof(icpgpu_result))
const char* icpgpu_multi_last_error(void);

/* ---- kernel-level entry points (used by the parity tests; same kernels as align) ------------- */
/* a2: idx[i], d2[i] = exact nearest neighbour of T*source[i] in target (lowest index on ties);
 * idx = -1, d2 = +inf when the target is empty or the point is non-finite. */
int icpgpu_nn(icpgpu_ctx* ctx, const float* T, int32_t* idx, float* d2);
/* a3+a4: over pairs of the last icpgpu_nn/align NN sweep with (double)d2 <= max_dist^2:
 * sums = {n, sum p(3), sum q(3), sum q p^T (9, row = q), sum d2}, p = T*source[i], q = target[idx[i]]. */
int icpgpu_reduce(icpgpu_ctx* ctx, const float* T, double max_dist, double sums[17]);
/* a5 (host): Umeyama without scaling from the 17 sums -> double[16] column-major. */
int icpgpu_solve(const double sums[17], double Tk[16]);
/* a6: out = T * source (w = 1), pcl::transformPointCloud (icp_odometer.cpp:205). */
int icpgpu_transform(icpgpu_ctx* ctx, const float* T, float* out_xyzw);

/* a11 (GICP mode): per-point regularised covariances U diag(1,1,1e-3) U^T of the 20 nearest neighbours
 * (pcl::GeneralizedIterativeClosestPoint::computeCovariances); out6 = n x {xx, xy, xz, yy, yz, zz}. */
int icpgpu_gicp_covariances(icpgpu_ctx* ctx, int of_target, double* out6);
/* ICPGPU_GICP_DEVICE=auto only: time GICP's two inner solvers on THIS box with the context's current source, target and parameters
 * (method GICP; a few alignments whose results are discarded) and keep the faster for the context's single alignments from now on.
 * *choice (nullable) = icpgpu_gicp_solver.  Never called implicitly: without it the fixed rule above holds, so two identical runs
 * take identical paths from their first alignment on.  Results cannot depend on the choice (same bits, tests/test_gpu_gicp.py). */
int icpgpu_calibrate(icpgpu_ctx* ctx, int* choice);
/* One evaluation of the QUADRATIC inner objective on the host (no device): sums = 75 double-double numbers as (hi, lo) pairs
 * (icpslam_amd/csrc/icp_gicp_quadratic.h: 60 A, 12 Bq, cq, m, sum d2), base16 = the guess (column-major float 4x4), x = (tx, ty,
 * tz, roll, pitch, yaw) -> f and its gradient as BFGS sees them.  A diagnostic entry: tests check the algebra with it. */
int icpgpu_gicp_quadratic_eval(const double* sums150, const float* base16, const double* x6, double* f, double* g6);
/* The device half of the same mode, alone: correspondences of T * source in the target (d2 < max_correspondence_distance^2), their
 * Mahalanobis matrices at T's rotation, and the 75 sums of the quadratic form as (hi, lo) pairs.  Diagnostic entry (tests). */
int icpgpu_gicp_quadratic_sums(icpgpu_ctx* ctx, const float* T, double* sums150);

/* ---- the step before the path: voxel-grid down-sampling -------------------- */
/* replaces IcpOdometer::voxelFilterCloud = pcl::VoxelGrid<PointXYZ>::filter with leaf (L, L, L)
 * (/root/reference/src/icpslam/icp_odometer.cpp:96-101,177; leaf 0.2 m in config/icpslam.yaml:14):
 * one output point per occupied cell = mean of its points, ascending cell-index order, pad = 1.0f;
 * non-finite points are skipped; if the cell index space overflows int32 the input is returned
 * unchanged (PCL's "leaf size is too small" behaviour). out_xyzw must hold n points. */
int icpgpu_voxel_grid(icpgpu_ctx* ctx, const float* xyzw, size_t n, float leaf, float* out_xyzw, size_t* n_out);
/* two-step form for callers that size their output by the result (pcl::VoxelGrid::filter resizes `output`): pass
 * out_xyzw = NULL above -- *n_out is the number of voxels, the filtered cloud stays in HBM -- then fetch it into a
 * buffer of `capacity` >= *n_out points.  Valid until the context's next voxel-filter call. */
int icpgpu_voxel_grid_fetch(icpgpu_ctx* ctx, float* out_xyzw, size_t capacity, size_t* n_out);
/* one-step form of the same (1.1): filter, and hand the result out as a VIEW -- *view_xyzw points at *n_out points in the context's
 * pinned staging buffer, valid until the context's next call.  The points are written there by a kernel queued behind the filter's
 * last one and arrive in front of the cell count the host waits for anyway: no second round trip, no copy engine
 * (pcl::VoxelGrid::filter(output): `output.points.assign(view, view + n)`).  The cloud also stays in HBM, and icpgpu_set_source
 * recognises the host copy, exactly as after icpgpu_voxel_grid. */
int icpgpu_voxel_grid_view(icpgpu_ctx* ctx, const float* xyzw, size_t n, float leaf, const float** view_xyzw, size_t* n_out);
/* the odometer's pre-step fused with setInputSource: upload, filter on the device, and make the
 * filtered cloud the source without a round trip to the host (icp_odometer.cpp:177 then :193). */
int icpgpu_set_source_voxel_filtered(icpgpu_ctx* ctx, const float* xyzw, size_t n, float leaf, size_t* n_out);

/* ---- the mapper's target: a one-point-per-voxel map and its "nn cloud"  -------- */
/* replaces OctreeMapper's pcl::octree::OctreePointCloudSearch map
 * (/root/reference/src/icpslam/octree_mapper.cpp:55-59 resetMap, :62-69 addPointsToMap,
 *  :72-90 approxNearestNeighbors, :133-172 refineTransformAndGrowMap; octree_resolution_ 0.5 m, :41).
 * The map belongs to the context and lives in HBM.
 *   reset        resetMap(): empty map with voxel size `resolution` (> 0).
 *   add_points   addPointsToMap(transformCloudToPoseFrame(cloud, pose)): p = pose * x (float, the a6
 *                contract); going through the points IN ORDER, p is appended to the map iff its voxel
 *                holds no point yet.  Voxels are the cells floor((p - origin) / resolution) (double) of
 *                the lattice whose origin is (first point ever added) - resolution -- PCL's octree
 *                bounding-box rule (first box p +- resolution / 2, widened to 2 voxels by getKeyBitSize).  Non-finite points are skipped.  pose NULL = identity.
 *   add_source   the same for the context's current source cloud (already in HBM).
 *   nn_target    approxNearestNeighbors + transformCloudToPoseFrame(.., raw_pose.inverse()) + setInputTarget:
 *                for every source point s (in order) the map point nearest to pose * s -- EXACT, lowest map
 *                index among equals, where PCL's approxNearestSearch is a heuristic -- moved by pose_inv,
 *                becomes the context's TARGET cloud (device to device).  Source points whose image is not
 *                finite are dropped, like the reference's "result_index < 0".  nn_out_xyzw (nullable) must
 *                hold n_source points; *n_nn = points in the nn cloud.  An empty map gives an empty target. */
/*   set_search   which neighbour nn_target collects: ICPGPU_MAP_SEARCH_EXACT (default, above) or ICPGPU_MAP_SEARCH_PCL_APPROX =
 *                what octree_mapper.cpp:84 literally calls, OctreePointCloudSearch::approxNearestSearch: from the root of
 *                PCL's octree (bounding box grown point by point as adoptBoundingBoxToPoint does) to the existing child
 *                whose voxel centre is nearest to the query (float squared distance, first child on ties), down to a leaf,
 *                whose point is returned -- a heuristic that misses the true neighbour for ~40 % of a scan's points.
 *                The exact search gives the better registration; this one restates the reference's own search (octree
 *                geometry per PCL 1.8's adoptBoundingBoxToPoint + getKeyBitSize; like every PCL restatement here it is
 *                unpinned against a PCL build). */
enum { ICPGPU_MAP_SEARCH_EXACT = 0, ICPGPU_MAP_SEARCH_PCL_APPROX = 1 };
int icpgpu_map_set_search(icpgpu_ctx* ctx, int mode);
int icpgpu_map_reset(icpgpu_ctx* ctx, double resolution);
int icpgpu_map_add_points(icpgpu_ctx* ctx, const float* xyzw, size_t n, const float* pose, size_t* n_added);
int icpgpu_map_add_source(icpgpu_ctx* ctx, const float* pose, size_t* n_added);
int icpgpu_map_size(icpgpu_ctx* ctx, size_t* n);
int icpgpu_map_get_points(icpgpu_ctx* ctx, float* out_xyzw, size_t capacity, size_t* n);
int icpgpu_map_nn_target(icpgpu_ctx* ctx, const float* pose, const float* pose_inv, float* nn_out_xyzw, size_t* n_nn);

/* ---- the data contract after the path: pose chain, keyframes, pose graph ---- */
/* SE(3) pose as the reference's Pose6DOF keeps it (/root/reference/include/utils/pose6DOF.h):
 * position + unit quaternion (x, y, z, w). Host-only arithmetic in double; no GPU involved. */
typedef struct {
  double pos[3];
  double quat[4]; /* x, y, z, w */
} icpgpu_pose;
typedef struct icpgpu_posegraph icpgpu_posegraph; /* opaque */

/* Pose6DOF(T): pose6DOF.cpp:185-190 (T = float[16] column-major as returned in icpgpu_result.T). */
int icpgpu_pose_from_matrix(const float* T, icpgpu_pose* out);
/* Pose6DOF::compose (operator+): pose6DOF.cpp:98-105.  Pose6DOF::inverse: pose6DOF.cpp:117-122. */
/* Pose6DOF::toTFTransform -> the Matrix4f pcl_ros::transformPointCloud applies (pose6DOF.cpp:254-259): column-major float */
int icpgpu_pose_to_matrix(const icpgpu_pose* p, float* T);
int icpgpu_pose_compose(const icpgpu_pose* a, const icpgpu_pose* b, icpgpu_pose* out);
int icpgpu_pose_inverse(const icpgpu_pose* a, icpgpu_pose* out);

/* Sequence bookkeeping of IcpOdometer::updateICPOdometry (icp_odometer.cpp:109-113) + the keyframe and
 * edge rules of IcpSlam::mainLoop / addNewKeyframe (icpslam.cpp:143-152, 70-89).
 *   keyframe_distance  : KFS_DIST_THRESH (icpslam.h:36, 0.3 m); < 0 = default
 *   information_diag6  : icp_information_matrix (config/icpslam.yaml:21); NULL = that default */
int icpgpu_posegraph_create(icpgpu_posegraph** out, double keyframe_distance, const double* information_diag6);
int icpgpu_posegraph_destroy(icpgpu_posegraph* g);
int icpgpu_posegraph_set_initial_pose(icpgpu_posegraph* g, const icpgpu_pose* p); /* IcpOdometer::setInitialPose */
/* one registration result per scan, in scan order. accepted = hasConverged() && fitness < 20
 * (icp_odometer.cpp:201); rejected scans leave the chain untouched. *keyframe_id = new keyframe or -1. */
int icpgpu_posegraph_push(icpgpu_posegraph* g, const float* T, int accepted, long* keyframe_id);
long icpgpu_posegraph_num_poses(const icpgpu_posegraph* g);
long icpgpu_posegraph_num_keyframes(const icpgpu_posegraph* g);
int icpgpu_posegraph_get_pose(const icpgpu_posegraph* g, long i, icpgpu_pose* out);
int icpgpu_posegraph_get_keyframe(const icpgpu_posegraph* g, long i, icpgpu_pose* out, long* scan_index);
/* edge measurement between keyframe new_kf and new_kf - 1: new^-1 (+) prev (icpslam.cpp:82). */
int icpgpu_posegraph_get_edge(const icpgpu_posegraph* g, long new_kf, icpgpu_pose* out);
/* g2o text file: VERTEX_SE3:QUAT id x y z qx qy qz qw / EDGE_SE3:QUAT from to x y z qx qy qz qw + the 21
 * upper-triangular information entries -- what PoseGraphG2O::addSe3Node/addSe3Edge would have built. */
int icpgpu_posegraph_write_g2o(const icpgpu_posegraph* g, const char* path);

/* ---- measurement ---------------------------------------------------------------------------- */
/* time one correspondence sweep in `every` (default 13, coprime with the reference's 10 / 30 iterations; 1 = every sweep).
 * Average kernel durations are *_ms / *_timed. */
int icpgpu_profile_set_sampling(icpgpu_ctx* ctx, int every);
int icpgpu_profile_reset(icpgpu_ctx* ctx);
int icpgpu_profile_get(icpgpu_ctx* ctx, icpgpu_profile* out);
/* stream handle (hipStream_t as void*) so a host can order its own work against the context. */
int icpgpu_get_stream(icpgpu_ctx* ctx, void** out_stream);
int icpgpu_synchronize(icpgpu_ctx* ctx);
/* Counting runs (bench.py's useful-flop figure; process-wide, one context at a time): with enable != 0 every grid
 * correspondence sweep (nn_quad_kernel) adds the number of target points it evaluates to a device counter;
 * icpgpu_count_candidates_read waits for the stream and returns the count since it was enabled (or last read), then
 * zeroes it.  Not for production: one atomic per wave. */
int icpgpu_count_candidates(icpgpu_ctx* ctx, int enable);
int icpgpu_count_candidates_read(icpgpu_ctx* ctx, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ICPGPU_H */
