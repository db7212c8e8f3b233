// icpgpu_batch.cpp -- icpgpu_align_batch: many independent scan pairs through one context (BASELINE configs 4 / 5).
#include "icp_ctx.h"


namespace icpgpu_impl {

// The buffers a batch worker will need for pairs of up to ns x nt points, allocated BEFORE the batch runs (by the threads that
// create the workers, side by side) instead of at the worker's first pair: a worker's first use is a dozen hipMallocs in the
// middle of a running batch, and with 64 workers dealt pairs dynamically some of them meet their first pair only in the second,
// third or fifth call of a process (traced in round 5: every such call took 17-20 ms instead of 12.5).
static int presize_batch_worker(icpgpu_ctx* w, size_t ns, size_t nt, size_t table_cells) {
  if (w->batch_presized_src >= ns && w->batch_presized_tgt >= nt && w->batch_presized_cells >= table_cells) return ICPGPU_OK;
  int rc = ICPGPU_OK;
  auto need = [&](DeviceBuf& b, size_t bytes) { if (!rc && bytes) rc = ensure(w, b, bytes); };
  if (w->src.buf.external) w->src.buf = DeviceBuf{};
  if (w->tgt.buf.external) w->tgt.buf = DeviceBuf{};
  need(w->src.buf, ns * sizeof(float4));
  need(w->tgt.buf, nt * sizeof(float4));
  need(w->keys, ns * sizeof(unsigned long long));
  need(w->prev.buf, ns * sizeof(unsigned int));
  need(w->partials, (size_t)std::max(grid_search_blocks((int)ns), kMaxReduceBlocks) * kReduceTerms * sizeof(double));
  GridIndex& G = w->grid;
  need(G.ints, (6 + kGridStatInts) * sizeof(int));
  need(G.cell_of_point, nt * sizeof(int));
  need(G.rank, nt * sizeof(int));
  need(G.sorted, nt * sizeof(float4));
  need(G.unmatched, (ns + 1) * sizeof(int));
  if (table_cells) {
    const size_t cells = std::min<size_t>(table_cells + table_cells / 4, (size_t)kMaxGridCells + 1);
    need(G.cell_start, cells * sizeof(int));
    need(G.block_sums, (cells / kScanItems + 2) * sizeof(int));
  }
  if (!rc) {
    w->batch_presized_src = std::max(w->batch_presized_src, ns);
    w->batch_presized_tgt = std::max(w->batch_presized_tgt, nt);
    w->batch_presized_cells = std::max(w->batch_presized_cells, table_cells);
  }
  return rc;
}

}  // namespace icpgpu_impl

extern "C" {

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
  else if (const char* w = ICPGPU_DEV_ENV("ICPGPU_BATCH_WORKERS")) t = (size_t)std::max(0, std::atoi(w));  // round-1 name
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
static int align_batch_impl(icpgpu_ctx* c, size_t n_pairs, const float* const* src, const size_t* n_src, const float* const* tgt,
                            const size_t* n_tgt, int want_fitness, icpgpu_result* results);

int icpgpu_align_batch(icpgpu_ctx* c, size_t n_pairs, const float* const* src, const size_t* n_src,
                       const float* const* tgt, const size_t* n_tgt, int want_fitness, icpgpu_result* results) {
  ENTER(c);
  if (n_pairs && (!src || !n_src || !tgt || !n_tgt || !results)) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  if (n_pairs == 0) return ICPGPU_OK;
  if (c->abi_result == sizeof(icpgpu_result)) return align_batch_impl(c, n_pairs, src, n_src, tgt, n_tgt, want_fitness, results);
  // the caller's icpgpu_result has another length (include/icpgpu.h, ABI rule): its array has ITS stride
  std::vector<icpgpu_result> own(n_pairs);
  const int rc = align_batch_impl(c, n_pairs, src, n_src, tgt, n_tgt, want_fitness, own.data());
  unsigned char* out = reinterpret_cast<unsigned char*>(results);
  for (size_t k = 0; k < n_pairs; ++k) {
    std::memset(out + k * c->abi_result, 0, c->abi_result);
    std::memcpy(out + k * c->abi_result, &own[k], std::min(c->abi_result, sizeof(icpgpu_result)));
  }
  return rc;
}

static int align_batch_impl(icpgpu_ctx* c, size_t n_pairs, const float* const* src, const size_t* n_src, const float* const* tgt,
                            const size_t* n_tgt, int want_fitness, icpgpu_result* results) {
  const bool gicp = c->params.method == ICPGPU_GICP;
  // Point-to-point batches run in LOCK-STEP (ICPGPU_BATCH_LOCKSTEP=0: the round-robin scheduler of round 2): a host thread
  // leads a group of `depth` pairs on ONE stream -- their index builds go through their host round trips together, and
  // every ICP iteration of the whole group is one search launch (pair = blockIdx.y, nn_quad_batch_kernel) + one final
  // reduction launch (17 x K workgroups) + K host solves, instead of K x (launch + reduce + poll): ~35 launches per K
  // pairs where there were ~35 per pair.  Two threads keep the GPU fed (one copies its next group in while the other's
  // group iterates); results are bit-identical to icpgpu_align's (same kernels' bodies, same workgroup -> point mapping).
  static const bool lockstep_on = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BATCH_LOCKSTEP"); return !e || std::atoi(e) != 0; }();
  const bool lockstep = lockstep_on && !gicp;
  // GICP: resumable runs leave the host free (two threads x four runs measured 1.72k pairs/s on 13k-point clouds, one x eight 1.6-1.7k,
  // eight x one 1.5k: profiles/r05_gicp_batch.txt); without the device solver every run is a blocking host loop and wants its own thread
  if (gicp) {
    const int grc = ensure_gicp_resources(c);
    if (grc) return grc;
  }
  const bool gicp_runs = gicp && (gicp_inner_quadratic(c) || (gicp_device_solver_mode() != 0 && c->gicp_device_ok));  // (quadratic runs need no device solver)
  size_t n_threads = batch_threads(c, gicp ? (gicp_runs ? 2 : 8) : 4), depth = 1;
  if (!gicp) {
    if (const char* v = std::getenv("ICPGPU_BATCH_DEPTH")) depth = (size_t)std::max(1, std::atoi(v));
    else depth = lockstep ? 8 : std::max<size_t>(2, (8 + n_threads - 1) / n_threads);
    if (lockstep) depth = std::min<size_t>(depth, (size_t)kBatchMax);
  } else {
    // GICP: kGicpRunsInFlight resumable runs over all threads (each keeps its solver's workgroups resident: eight of them, one
    // per XCD, is what the chip holds with room for everybody's searches -- kMaxServerWorkers); ICPGPU_BATCH_DEPTH = runs per thread
    if (const char* v = std::getenv("ICPGPU_BATCH_DEPTH")) depth = (size_t)std::max(1, std::atoi(v));
    else if (gicp_runs) depth = std::max<size_t>(1, (16 + n_threads - 1) / n_threads);  // 16 runs in flight: their solvers leave in combined launches (below)
    else depth = std::max<size_t>(1, (kMaxServerWorkers + n_threads - 1) / n_threads);
  }
  n_threads = std::min(n_threads, n_pairs);
  depth = std::min(depth, (n_pairs + n_threads - 1) / n_threads);
  // lock-step: groups in flight on the GPU = n_threads x groups_per_thread, eight by default whatever the CPUs (a 50k-point
  // group's sweep leaves the chip idle in its tail; measured at config 4's shape, 64 pairs: profiles/r05_batch_groups.txt).
  // ICPGPU_BATCH_GROUPS sets the total.
  size_t groups_per_thread = 1;
  if (lockstep) {
    size_t total = 8;
    if (const char* v = std::getenv("ICPGPU_BATCH_GROUPS")) total = (size_t)std::max(1, std::atoi(v));
    total = std::min(total, (n_pairs + depth - 1) / depth);
    n_threads = std::min(n_threads, std::max<size_t>(1, total));
    groups_per_thread = std::max<size_t>(1, (total + n_threads - 1) / n_threads);
  }
  const size_t per_thread = groups_per_thread * depth;
  const size_t n_ctx = n_threads * per_thread;
  // Worker contexts are created on first use, by the batch threads in parallel (a context is a stream, two hundred events and a
  // dozen pinned / fine-grained / device allocations: ~1 ms, and a batch wants up to 64 of them -- created one after the other by
  // the calling thread they were most of the first call's time)
  double create_ms = 0.0;
  if (c->workers.size() < n_ctx) {
    const auto t_create = std::chrono::steady_clock::now();
    const size_t have = c->workers.size();
    c->workers.resize(n_ctx, nullptr);
    std::vector<int> create_rc(n_ctx, ICPGPU_OK);
    std::vector<std::string> create_msg(n_ctx);  // (icpgpu_last_error(nullptr) is per thread: taken where it was set)
    std::vector<std::thread> makers;
    const size_t n_makers = std::min<size_t>(std::max<size_t>(1, n_threads), n_ctx - have);
    for (size_t m = 0; m < n_makers; ++m)
      makers.emplace_back([&, m] {
        for (size_t i = have + m; i < n_ctx; i += n_makers) {
          icpgpu_ctx* w = nullptr;
          create_rc[i] = create_context(&w, c->device, /*with_stream=*/!(lockstep || gicp_runs));  // (lock-step workers run on their group's stream, GICP runs on their thread's)
          if (create_rc[i] == ICPGPU_OK) w->shared_table_cells = &c->batch_table_cells;
          else create_msg[i] = icpgpu_last_error(nullptr);
          c->workers[i] = w;
        }
      });
    for (auto& th : makers) th.join();
    for (size_t i = have; i < n_ctx; ++i)
      if (create_rc[i] != ICPGPU_OK) {
        const int rc = create_rc[i];
        for (size_t k = have; k < n_ctx; ++k)
          if (c->workers[k]) icpgpu_destroy(c->workers[k]);
        c->workers.resize(have);
        return fail(c, rc, "align_batch: worker context: %s", create_msg[i].c_str());
      }
    create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
  }
  if (lockstep) {  // every worker's buffers for this batch's largest clouds (nothing to do once they are there)
    size_t ns_max = 0, nt_max = 0;
    for (size_t k = 0; k < n_pairs; ++k) {
      ns_max = std::max(ns_max, n_src[k]);
      nt_max = std::max(nt_max, n_tgt[k]);
    }
    const size_t cells = c->batch_table_cells.load(std::memory_order_relaxed);
    bool all_there = true;
    for (size_t i = 0; i < n_ctx && all_there; ++i)
      all_there = c->workers[i]->batch_presized_src >= ns_max && c->workers[i]->batch_presized_tgt >= nt_max && c->workers[i]->batch_presized_cells >= cells;
    if (!all_there) {
      const auto t_pre = std::chrono::steady_clock::now();
      std::vector<int> pre_rc(n_ctx, ICPGPU_OK);
      std::vector<std::thread> makers;
      const size_t n_makers = std::min<size_t>(std::max<size_t>(1, n_threads), n_ctx);
      for (size_t m = 0; m < n_makers; ++m)
        makers.emplace_back([&, m] {
          if (hipSetDevice(c->device) != hipSuccess) {  // (reported through the workers this maker was responsible for)
            for (size_t i = m; i < n_ctx; i += n_makers) pre_rc[i] = fail(c->workers[i], ICPGPU_ERR_HIP, "hipSetDevice failed in a batch maker thread");
            return;
          }
          for (size_t i = m; i < n_ctx; i += n_makers) pre_rc[i] = presize_batch_worker(c->workers[i], ns_max, nt_max, cells);
        });
      for (auto& th : makers) th.join();
      for (size_t i = 0; i < n_ctx; ++i)
        if (pre_rc[i] != ICPGPU_OK) return fail(c, pre_rc[i], "align_batch: worker buffers: %s", c->workers[i]->err.c_str());
      create_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pre).count();
    }
  }
  // The groups' streams are created HERE, one after the other: the runtime spreads streams over its (four) hardware queues in
  // creation order, and kernels of streams that share a hardware queue do not overlap.  Until round 5 a group ran on its lead
  // worker's stream -- the (g x depth)-th stream the context created -- so that with a depth of 8 (or 4) EVERY group sat on the
  // same hardware queue and the groups' sweeps ran one after the other: 4.1-4.5k pairs/s at config 4's shape where depths of
  // 5 or 7 reached 5.1-5.5k (rocprofv3 kernel trace: 2 queues in use, mostly one search kernel at a time, against 4 queues and
  // 2-4 kernels; profiles/r05_batch_groups.txt).
  // (hipStreamCreate is 3-4 ms apiece on this runtime when called in a row -- scripts/probes/create_probe.cpp -- and well under a
  //  millisecond each from several threads at once: the missing streams are created side by side, then appended in order)
  // (GICP runs: four streams per thread, created in a row = one per hardware queue: the combined solver launches on the first,
  //  the runs' short kernels spread over the others)
  const size_t streams_wanted = lockstep ? n_threads * groups_per_thread : (gicp_runs ? 4 * n_threads : 0);
  if (c->group_streams.size() < streams_wanted) {
    const size_t have = c->group_streams.size(), want = streams_wanted;
    std::vector<hipStream_t> fresh(want - have, nullptr);
    std::vector<hipError_t> fresh_rc(want - have, hipSuccess);
    std::vector<std::thread> makers;
    for (size_t k = 0; k < fresh.size(); ++k)
      makers.emplace_back([&, k] {
        fresh_rc[k] = hipSetDevice(c->device);
        if (fresh_rc[k] == hipSuccess) fresh_rc[k] = hipStreamCreateWithFlags(&fresh[k], hipStreamNonBlocking);
      });
    for (auto& th : makers) th.join();
    for (size_t k = 0; k < fresh.size(); ++k)
      if (fresh_rc[k] != hipSuccess) {
        for (hipStream_t st : fresh)
          if (st) (void)hipStreamDestroy(st);
        return fail(c, ICPGPU_ERR_HIP, "align_batch: hipStreamCreate: %s", hipGetErrorString(fresh_rc[k]));
      }
    for (hipStream_t st : fresh) c->group_streams.push_back(st);
  }
  const auto t_call = std::chrono::steady_clock::now();
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
    icpgpu_ctx* const* ws = &c->workers[t * per_thread];
    for (size_t s = 0; s < per_thread; ++s) {
      ws[s]->params = c->params;
      ws[s]->nn_variant = c->nn_variant;
      // Every GICP worker's BFGS runs keep its workgroups resident and a host thread spinning.  8 workers fit the chip
      // with room for everybody's searches; 16 were measured 5x SLOWER than single launches (servers wait for slots other
      // servers hold until their 50 ms patience runs out).
      ws[s]->gicp_server_allowed = n_threads * (gicp ? depth : 1) <= 2 * kMaxServerWorkers;
      // ... and each worker's evaluations get their share of the ~512 workgroups of that size the chip holds at once (64
      // apiece for 8 workers, as measured in round 1; a lone alignment uses up to 256)
      ws[s]->gicp_blocks_most = std::max(16, std::min(kGicpDirectBlocks, 512 / (int)std::max<size_t>(1, n_threads * (gicp ? depth : 1))));
    }
    if (!lockstep && !gicp_runs)  // these paths run every worker on its own stream
      for (size_t s = 0; s < per_thread; ++s)
        if (ensure_stream(ws[s])) return failed(ICPGPU_ERR_HIP, 0, ws[s]);
    if (gicp) {
      // GICP: `depth` resumable runs per thread, round-robin (GicpRun, icpgpu_gicp.cpp): the thread queues a run's next stage when
      // its awaited result has arrived and looks at the others meanwhile -- until round 5 it was ONE blocking alignment per
      // thread, its host spinning through every inner minimisation.  A run that cannot be resumable (no device solver on its
      // context, or the solver gave up) is one blocking alignment inside its step.
      // With the device solver (gicp_runs): the solvers of the runs that are ready go out TOGETHER, one gicp_solve_batch_kernel
      // launch on the thread's solve stream behind the runs' events -- a resident solver kernel per run would hold a hardware
      // queue each, four registrations at most solving at any time.  The runs' short kernels (index builds, covariances, searches,
      // fitness) go to three work streams, created right behind the solve stream and therefore on the three OTHER hardware queues:
      // nothing short ever waits behind a solver.  (One work stream for all runs was tried first: every run's next step then
      // queued behind every other run's covariance pass -- 0.5-1.1k pairs/s instead of 1.7k.)
      std::vector<GicpRun> gruns(depth);
      std::vector<size_t> pair_of(depth, 0);
      bool exhausted = false;
      // Two solve streams, two work streams.  A solve stream takes a new launch only when the runs of its last one have all
      // answered (launches on one stream run one after the other: a launch per ready run, issued at once, put every run behind
      // every other -- 0.57k pairs/s); meanwhile the runs that become ready collect and leave together.
      hipStream_t work_stream[2] = {nullptr, nullptr}, solve_stream[2] = {nullptr, nullptr};
      std::vector<hipStream_t> own(depth, nullptr);
      std::vector<int> solving_on(depth, -1);  // which solve stream a run's solver is on (-1: none)
      int blocks_on[2] = {0, 0};               // solver workgroups the launch in flight on each solve stream holds
      struct RestoreStreams {
        std::vector<hipStream_t>& own; icpgpu_ctx* const* ws; hipStream_t* w; hipStream_t* s2;
        ~RestoreStreams() {
          if (!s2[0]) return;
          for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(s2[k]);
          for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(w[k]);
          for (size_t i = 0; i < own.size(); ++i) ws[i]->stream = own[i];
        }
      } restore_streams{own, ws, work_stream, solve_stream};
      if (gicp_runs) {
        for (int k = 0; k < 2; ++k) solve_stream[k] = c->group_streams[4 * t + k];
        for (int k = 0; k < 2; ++k) work_stream[k] = c->group_streams[4 * t + 2 + k];
        for (size_t s2 = 0; s2 < depth; ++s2) {
          own[s2] = ws[s2]->stream;
          ws[s2]->stream = work_stream[s2 % 2];
        }
      }
      for (;;) {
        bool progressed = false;
        size_t in_flight = 0;
        for (size_t s2 = 0; s2 < depth; ++s2) {
          GicpRun& r = gruns[s2];
          icpgpu_ctx* w = ws[s2];
          if (r.phase == GicpRun::Idle || r.phase == GicpRun::Done) {
            if (exhausted || abort.load()) continue;
            const size_t k = next.fetch_add(1);
            if (k >= n_pairs) {
              exhausted = true;
              continue;
            }
            pair_of[s2] = k;
            int rc = load_pair(w, k, /*sync=*/false);
            if (!rc) rc = gicp_run_begin(w, r, want_fitness, &results[k], /*combine=*/gicp_runs);
            if (rc) return failed(rc, k, w);
            progressed = true;
            if (r.phase != GicpRun::Done) ++in_flight;
          } else {
            const int st = gicp_run_step(w, r);
            if (st < 0) return failed(st, pair_of[s2], w);
            progressed = progressed || st > 0;
            if (r.phase != GicpRun::Done) ++in_flight;
          }
        }
        if (in_flight == 0 && (exhausted || abort.load())) return;
        for (int ss = 0; gicp_runs && ss < 2; ++ss) {  // the solvers of every run that is ready: one launch, on a solve stream that is idle
          bool busy = false, other_busy = false;
          for (size_t s2 = 0; s2 < depth; ++s2) {
            if (solving_on[s2] >= 0 && gruns[s2].phase != GicpRun::Solve) solving_on[s2] = -1;  // (answered, or gave up)
            busy = busy || solving_on[s2] == ss;
            other_busy = other_busy || solving_on[s2] == 1 - ss;
          }
          if (busy) continue;
          blocks_on[ss] = 0;
          if (!other_busy) blocks_on[1 - ss] = 0;
          GicpSolveItem items[kGicpSolveBatchMax];
          size_t who[kGicpSolveBatchMax];
          int n_ready = 0, blocks_sum = 0;
          // (a solver workgroup takes a CU's one-wave-per-SIMD slot: the workgroups resident over BOTH solve streams of ALL threads
          //  together stay below the chip's 256 CUs, or a run's workgroups would spin waiting for peers that have no CU yet.  Until
          //  round 6 the cap was applied per launch, and two launches of a thread could hold ~2 x 240 / n_threads between them.)
          const int blocks_cap = std::max(1, 240 / (int)n_threads) - blocks_on[1 - ss];
          for (size_t s2 = 0; s2 < depth && n_ready < kGicpSolveBatchMax; ++s2)
            if (gruns[s2].phase == GicpRun::WantSolve && ((n_ready == 0 && !other_busy) || blocks_sum + gruns[s2].item.blocks <= blocks_cap)) {
              items[n_ready] = gruns[s2].item;
              blocks_sum += gruns[s2].item.blocks;
              who[n_ready++] = s2;
            }
          if (n_ready) {
            hipError_t e = hipSuccess;
            for (int k = 0; k < n_ready && e == hipSuccess; ++k) {  // behind each run's own search and Mahalanobis kernels
              icpgpu_ctx* w = ws[who[k]];
              e = hipEventRecord(w->ev[0], w->stream);
              if (e == hipSuccess) e = hipStreamWaitEvent(solve_stream[ss], w->ev[0], 0);
            }
            if (e == hipSuccess) e = launch_gicp_solve_batch(items, n_ready, 20, 1e-2, solve_stream[ss]);
            if (e != hipSuccess) {
              fail(ws[who[0]], ICPGPU_ERR_HIP, "GICP batch: solver launch: %s", hipGetErrorString(e));
              return failed(ICPGPU_ERR_HIP, pair_of[who[0]], ws[who[0]]);
            }
            for (int k = 0; k < n_ready; ++k) {
              gicp_run_solver_launched(ws[who[k]], gruns[who[k]], solve_stream[ss]);
              solving_on[who[k]] = ss;
            }
            blocks_on[ss] = blocks_sum;
            progressed = true;
          }
        }
#if defined(__x86_64__)
        if (!progressed) __builtin_ia32_pause();
#endif
      }
    }
    const double timeout_ms = wait_timeout_ms();
    if (lockstep) {
      // ---- lock-step groups ----------------------------------------------------------------------------------------
      // A host thread leads `groups_per_thread` groups, each `depth` pairs on a stream of its own, as NON-BLOCKING state
      // machines (fill -> index builds -> iterate): while one group's sweep is in flight the thread copies the next pair of
      // another group in, takes that group's builds through a host round trip, or does its solves.  The number of groups in
      // flight on the GPU is therefore decoupled from the number of CPUs the process may use (round 4: one group per thread,
      // four threads at most -- 4.0k pairs/s at config 4's shape where eight such entries sharing the GPU reached 5.2k: the
      // sweeps of one group leave the chip idle in their tails, more groups fill them).
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
        unsigned int* prev = nullptr;
      };
      struct Group {
        enum Phase { Fill, Build, Iterate, Retired } phase = Fill;
        icpgpu_ctx* const* ws = nullptr;
        icpgpu_ctx* lead = nullptr;
        hipStream_t gstream = nullptr;  // copies, index builds, the table: the lead's own stream
        hipStream_t sstream = nullptr;  // the sweeps (search + reduction + the fitness tail): the same stream (see below)
        std::vector<hipStream_t> own;
        std::vector<Slot> slots;
        std::vector<BatchPair> table;
        size_t n_slots = 0, live = 0, cap = 0;  // cap: pairs the current fill takes (the thread's first one is small: see below)
        unsigned idle_spins = 0, step_counter = 0;
        int timed_pairs = 0;  // pairs of the launch whose events are outstanding
        size_t index = 0;     // the group's number among all groups of the job
        bool filled_once = false;
        unsigned long long sync_seq = 0;  // Build: the number the group's posted marker carries once its stage has drained
        std::chrono::steady_clock::time_point bt0, bt1, bt2;
        ~Group() {  // whatever way the thread leaves: the workers get their own streams back, nothing of the group in flight
          if (sstream) (void)hipStreamSynchronize(sstream);
          if (gstream) (void)hipStreamSynchronize(gstream);
          for (size_t i = 0; i < own.size(); ++i) ws[i]->stream = own[i];
        }
      };
      static const bool stagger_on = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BATCH_STAGGER"); return !e || std::atoi(e) != 0; }();
      const size_t total_groups = n_threads * groups_per_thread;
      std::vector<Group> groups(groups_per_thread);
      for (size_t gi = 0; gi < groups_per_thread; ++gi) {
        Group& G = groups[gi];
        G.ws = ws + gi * depth;
        G.index = gi * n_threads + t;  // (a thread's groups are far apart: its first one starts early, its last one late)
        G.lead = G.ws[0];
        G.gstream = c->group_streams[G.index];
        // (Measured and dropped in round 5: the sweeps on a stream of the LOWEST priority, so that the small kernels of another group's
        // index build -- 1-5 ms per group on a busy chip, 0.4 ms on an idle one -- are dispatched ahead of the waiting sweep
        // workgroups: 15-25 % SLOWER in every configuration, copies stalling for milliseconds; profiles/r05_batch_groups.txt.)
        G.sstream = G.gstream;
        G.slots.resize(depth);
        G.table.resize(depth);
        G.own.resize(depth);
        for (size_t s2 = 0; s2 < depth; ++s2) {
          G.slots[s2].w = G.ws[s2];
          G.own[s2] = G.ws[s2]->stream;
          G.ws[s2]->stream = G.gstream;  // one queue for the group: builds, sweeps and fitness sweeps are ordered by it
        }
        if (ensure(G.lead, G.lead->batch_table, depth * sizeof(BatchPair))) return failed(ICPGPU_ERR_OOM, 0, G.lead);
      }
      static const bool bt_on = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BATCH_TIMING"); return e && std::atoi(e) != 0; }();
      static const bool bt_trace = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BATCH_TRACE"); return e && std::atoi(e) != 0; }();
      double bt_fill = 0, bt_build = 0, bt_iter = 0;
      size_t bt_groups = 0, bt_steps = 0;
      struct BtPrint {
        const bool& on; double &f, &b, &i; size_t &g, &st; size_t t;
        ~BtPrint() { if (on && g) fprintf(stderr, "[icpgpu] batch thread %zu: %zu groups, %zu steps; per group: fill (H2D) %.3f ms, index builds %.3f ms, iterations + fitness %.3f ms\n", t, g, st, f / g, b / g, i / g); }
      } bt_print{bt_on, bt_fill, bt_build, bt_iter, bt_groups, bt_steps, t};

      // One non-blocking step of a group.  1: something moved, 0: nothing did, 2: the group is retired (no pairs left), < 0: failed
      // (the thread's error is set).
      auto group_step = [&](Group& G) -> int {
        icpgpu_ctx* lead = G.lead;
        hipStream_t gstream = G.gstream;
        std::vector<Slot>& slots = G.slots;
        std::vector<BatchPair>& table = G.table;
        size_t& n_slots = G.n_slots;
        auto bail = [&](int rc, size_t k, icpgpu_ctx* w) { failed(rc, k, w); return -1; };
        // queue a marker behind everything the group's stream holds: the stage has drained once the mailbox carries its number
        auto post_marker = [&]() -> int {
          G.sync_seq = ++lead->post_seq;
          if (launch_post_ints(static_cast<const int*>(lead->batch_table.ptr), 1, lead->h_post_dev, wire_seq(lead, G.sync_seq), gstream) != hipSuccess) {
            fail(lead, ICPGPU_ERR_HIP, "lock-step batch: marker launch");
            return bail(ICPGPU_ERR_HIP, slots[0].pair, lead);
          }
          return 0;
        };
        if (G.phase == Group::Fill) {
          // (1) fill the group, ONE pair per step (a pair's host -> device copies occupy the thread for ~0.2 ms: the other
          // groups' mailboxes are looked at in between)
          if (n_slots == 0) {
            G.bt0 = std::chrono::steady_clock::now();
            // Ramp-up: groups that start together stay in phase -- all copying, then all building, then all sweeping, the chip idle
            // through the first two -- so the FIRST fill of the job's g-th group takes (g + 1) / n of a full group: the small ones
            // sweep (short steps, but on an otherwise idle chip) while the larger ones are still being copied in, and from then on
            // the groups finish and refill at different times.
            G.cap = depth;
            if (!G.filled_once && stagger_on) G.cap = std::max<size_t>(1, (depth * (G.index + 1) + total_groups - 1) / total_groups);
            G.filled_once = true;
          }
          bool full = n_slots == G.cap;
          if (!full && !abort.load()) {
            const size_t k = next.fetch_add(1);
            if (k < n_pairs) {
              Slot& sl = slots[n_slots];
              sl.pair = k;
              sl.lock = sl.wants = sl.wants_fit = sl.waiting = false;
              int rc = load_pair(sl.w, k, /*sync=*/false);
              if (!rc) rc = p2p_prepare(sl.w, sl.run, nullptr, nullptr, want_fitness, &results[k]);
              if (rc) return bail(rc, k, sl.w);
              ++n_slots;
              if (n_slots < G.cap) return 1;
            }
            full = true;  // (or no pair left: the group goes with what it has)
          }
          if (n_slots == 0 || abort.load()) {
            G.phase = Group::Retired;
            return 2;
          }
          G.bt1 = std::chrono::steady_clock::now();
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
            if (rc) return bail(rc, sl.pair, w);
          }
          if (post_marker()) return -1;
          G.phase = Group::Build;
          G.idle_spins = 0;
          return 1;
        }
        if (G.phase == Group::Build) {
          bool pending = false;
          for (size_t i = 0; i < n_slots; ++i) pending = pending || slots[i].gb.state != GridBuild::Done;
          if (pending) {
            // the stage's read-backs (hipMemcpyAsync into pinned memory, queued by the builds) have landed once the marker queued
            // behind them has: polled in host memory, no runtime call and no sleeping wait
            if ((lead->h_post[1] >> 24) < G.sync_seq) {
              if ((++G.idle_spins & 0xFFFu) == 0) {
                const hipError_t q = hipStreamQuery(gstream);
                if (q != hipSuccess && q != hipErrorNotReady) {
                  fail(slots[0].w, ICPGPU_ERR_HIP, "lock-step batch: %s", hipGetErrorString(q));
                  return bail(ICPGPU_ERR_HIP, slots[0].pair, slots[0].w);
                }
                if (q == hipSuccess && (lead->h_post[1] >> 24) < G.sync_seq) {  // drained without the marker: its pair was torn or lost -- fall back to the stream's word
                  std::atomic_thread_fence(std::memory_order_acquire);
                } else if (q == hipErrorNotReady) {
                  return 0;
                }
              } else {
                return 0;
              }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            G.idle_spins = 0;
            for (size_t i = 0; i < n_slots; ++i)
              if (slots[i].gb.state != GridBuild::Done) {
                const int rc = gb_advance(slots[i].w, slots[i].gb);
                if (rc) return bail(rc, slots[i].pair, slots[i].w);
              }
            pending = false;
            for (size_t i = 0; i < n_slots; ++i) pending = pending || slots[i].gb.state != GridBuild::Done;
            if (pending) {
              if (post_marker()) return -1;
              return 1;
            }
          }
          G.bt2 = std::chrono::steady_clock::now();
          // (3) who can iterate in lock-step; the others start their own first sweep
          G.live = 0;
          for (size_t i = 0; i < n_slots; ++i) {
            Slot& sl = slots[i];
            if (sl.run.phase == P2PRun::Done) continue;
            icpgpu_ctx* w = sl.w;
            ++G.live;
            const int n_s = (int)w->src.n;
            const int flags = grid_flags(w->grid, false);
            sl.lock = grid_ready(w) && n_s > 0 && w->src.n < kOrderSourceMin && source_order_mode() != 1 &&
                      grid_search_batchable(n_s, flags) && sl.run.thr <= w->grid.cutoff * w->grid.cutoff;
            sl.run.phase = P2PRun::Iterating;
            sl.run.t_issue = std::chrono::steady_clock::now();
            if (!sl.lock) {
              int rc = ensure_source_order(w, sl.run.thr);
              if (!rc) rc = sweep_issue(w, to_xform(sl.run.final_T), sl.run.thr, false, sl.run.ticket);
              if (rc) return bail(rc, sl.pair, w);
              continue;
            }
            if (w->src_grid.version != w->src_version) w->src_grid.built = w->src_grid.usable = false;
            const int blocks = grid_search_blocks(n_s);
            int rc = ensure(w, w->partials, (size_t)blocks * kReduceTerms * sizeof(double));
            bool use_prev = false;
            if (!rc) rc = prev_neighbours(w, w->grid, w->src.data(), n_s, flags, sl.prev, use_prev);  // (allocates; the first sweep is cold)
            if (rc) return bail(rc, sl.pair, w);
            w->prev.valid = w->tile_seed.valid = false;
            sl.pack = (flags & kGridPackShortRows) != 0;
            if (!rc) rc = ensure(w, w->keys, (size_t)n_s * sizeof(unsigned long long));
            if (!rc) rc = ensure(w, w->grid.unmatched, (size_t)(n_s + 1) * sizeof(int));
            if (rc) return bail(rc, sl.pair, w);
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
            return bail(ICPGPU_ERR_HIP, slots[0].pair, lead);
          }
          // the sweeps' stream takes over behind everything the builds' stream holds (the last build stage and the table are queued, not done)
          if (G.sstream != gstream &&
              (hipEventRecord(lead->ev[0], gstream) != hipSuccess || hipStreamWaitEvent(G.sstream, lead->ev[0], 0) != hipSuccess)) {
            fail(lead, ICPGPU_ERR_HIP, "lock-step batch: stream hand-over");
            return bail(ICPGPU_ERR_HIP, slots[0].pair, lead);
          }
          G.phase = Group::Iterate;
          G.idle_spins = 0;
          return 1;
        }
        // (4) iterate: one search launch + one reduction launch per step and row-walk class for the whole group
        if (G.live > 0) {
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
              unsigned int* buf = nullptr;
              if (prev_neighbours(w, w->grid, w->src.data(), (int)w->src.n, grid_flags(w->grid, false), buf, use_prev) || buf != sl.prev) {
                fail(w, ICPGPU_ERR_HIP, "lock-step batch: previous-neighbour buffer moved");
                return bail(ICPGPU_ERR_HIP, sl.pair, w);
              }
              step.T[n_act] = to_xform(sl.run.final_T);
              step.seq[n_act] = wire_seq(w, ++w->sums_seq);
              step.slot[n_act] = (unsigned char)i;
              if (use_prev) step.use_prev_mask |= 1u << n_act;
              max_blocks = std::max(max_blocks, table[i].blocks);
              sl.run.ticket = SweepTicket{};
              sl.run.ticket.seq = w->sums_seq;
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
                if (hipMemsetAsync(table[i].unmatched_count, 0, sizeof(int), G.sstream) != hipSuccess) {
                  fail(w, ICPGPU_ERR_HIP, "lock-step batch: memset");
                  return bail(ICPGPU_ERR_HIP, sl.pair, w);
                }
              }
              ++n_act;
            }
            if (n_act == 0) continue;
            bt_steps += 1;
            // kernel timing, sampled like the single-pair path's: a launch's HIP-event time / its pairs = one pair's sweep
            auto take_timing = [&](bool wait) {
              if (!G.timed_pairs) return;
              if (wait) (void)hipEventSynchronize(lead->ev[3]);
              else if (hipEventQuery(lead->ev[3]) != hipSuccess) return;
              float ms = 0.f;
              if (hipEventElapsedTime(&ms, lead->ev[2], lead->ev[3]) == hipSuccess) {
                lead->prof.grid_ms += (double)ms;  // (summed over the launch's pairs: grid_ms / grid_timed stays "per pair and sweep")
                lead->prof.grid_timed += (uint64_t)G.timed_pairs;
              }
              G.timed_pairs = 0;
            };
            take_timing(false);
            const bool timed = kind == 0 && G.timed_pairs == 0 && (G.step_counter++ % 5u) == 0;
            if (timed) (void)hipEventRecord(lead->ev[2], G.sstream);
            const BatchPair* d_table = static_cast<const BatchPair*>(lead->batch_table.ptr);
            hipError_t e = launch_nn_grid_search_batch(d_table, step, n_act, max_blocks, group_pack, kind == 1, G.sstream);
            if (timed) {
              (void)hipEventRecord(lead->ev[3], G.sstream);
              G.timed_pairs = n_act;
            }
            if (e == hipSuccess && kind == 0) e = launch_reduce_final_batch(d_table, step, n_act, G.sstream);
            if (e != hipSuccess) {
              fail(lead, ICPGPU_ERR_HIP, "lock-step batch launch: %s", hipGetErrorString(e));
              return bail(ICPGPU_ERR_HIP, slots[step.slot[0]].pair, lead);
            }
            if (kind == 1) {
              // per pair: the few points the grid left unmatched (completed on the device), then the keys-path reduction into
              // the pair's mailbox -- three small launches each, queued without waiting
              for (int a2 = 0; a2 < n_act; ++a2) {
                Slot& sl = slots[step.slot[a2]];
                icpgpu_ctx* w = sl.w;
                const BatchPair& bp = table[step.slot[a2]];
                hipError_t e2 = launch_nn_brute_few(w->src.data(), bp.unmatched, bp.unmatched_count, 0, w->tgt.data(), (int)w->tgt.n,
                                                    step.T[a2], bp.keys, reinterpret_cast<int*>(w->h_sums_dev + 20), G.sstream);
                int rc2 = e2 == hipSuccess ? ensure(w, w->partials, (size_t)kMaxReduceBlocks * kReduceTerms * sizeof(double)) : ICPGPU_ERR_HIP;
                if (!rc2 && launch_reduce(w->src.data(), (int)w->src.n, w->tgt.data(), bp.keys, step.T[a2], FLT_MAX,
                                          static_cast<double*>(w->partials.ptr), w->h_sums_dev, w->h_flags_dev, step.seq[a2], G.sstream) != hipSuccess)
                  rc2 = ICPGPU_ERR_HIP;
                if (rc2) {
                  fail(w, rc2, "lock-step batch: fitness sweep");
                  return bail(rc2, sl.pair, w);
                }
              }
            }
          }
          // take everything that has come back
          bool progressed = false;
          for (size_t i = 0; i < n_slots; ++i) {
            Slot& sl = slots[i];
            if (sl.run.phase != P2PRun::Iterating && sl.run.phase != P2PRun::Fitness) continue;
            if (sl.lock && !sl.waiting) continue;  // its sweep has not been launched yet
            if (!sweep_ready(sl.w, sl.run.ticket)) continue;
            sl.waiting = false;
            int deferred = 0;
            const int rc = p2p_advance(sl.w, sl.run, (sl.lock && sl.run.phase == P2PRun::Iterating) ? &deferred : nullptr);
            if (rc) return bail(rc, sl.pair, sl.w);
            sl.wants = deferred == 1;
            sl.wants_fit = deferred == 2;
            if (sl.run.phase == P2PRun::Done) --G.live;
            progressed = true;
          }
          if (progressed) {
            G.idle_spins = 0;
            if (G.live > 0) return 1;
          } else {
            if ((++G.idle_spins & 0x3FFu) == 0) {  // nothing moved for a while: a faulted or hung kernel must not keep us here
              const auto now = std::chrono::steady_clock::now();
              hipError_t q = hipStreamQuery(G.sstream);
              if (q == hipSuccess || q == hipErrorNotReady) {
                const hipError_t q2 = hipStreamQuery(gstream);
                if (q2 != hipSuccess && q2 != hipErrorNotReady) q = q2;
              }
              for (size_t i = 0; i < n_slots; ++i) {
                Slot& sl = slots[i];
                if (sl.run.phase != P2PRun::Iterating && sl.run.phase != P2PRun::Fitness) continue;
                if (q != hipSuccess && q != hipErrorNotReady) {
                  fail(sl.w, ICPGPU_ERR_HIP, "HIP error while waiting for a reduction: %s", hipGetErrorString(q));
                  return bail(ICPGPU_ERR_HIP, sl.pair, sl.w);
                }
                if (std::chrono::duration<double, std::milli>(now - sl.run.t_issue).count() > timeout_ms) {
                  fail(sl.w, ICPGPU_ERR_HIP, "timed out after %.0f ms waiting for a kernel's result (hung kernel?)", timeout_ms);
                  return bail(ICPGPU_ERR_HIP, sl.pair, sl.w);
                }
              }
            }
            return 0;
          }
        }
        // the group is finished
        if (G.timed_pairs) {  // (its last timed launch is, too)
          float ms = 0.f;
          if (hipEventSynchronize(lead->ev[3]) == hipSuccess && hipEventElapsedTime(&ms, lead->ev[2], lead->ev[3]) == hipSuccess) {
            lead->prof.grid_ms += (double)ms;
            lead->prof.grid_timed += (uint64_t)G.timed_pairs;
          }
          G.timed_pairs = 0;
        }
        if (bt_trace) {  // development flavour, ICPGPU_BATCH_TRACE=1: the round's phase boundaries, ms after the call started
          const auto ms = [&](std::chrono::steady_clock::time_point tp) { return std::chrono::duration<double, std::milli>(tp - t_call).count(); };
          fprintf(stderr, "[icpgpu] batch trace: thread %zu group %zu: %zu pairs | fill %.2f .. %.2f | builds .. %.2f | sweeps .. %.2f\n", t, G.index, n_slots,
                  ms(G.bt0), ms(G.bt1), ms(G.bt2), ms(std::chrono::steady_clock::now()));
        }
        if (bt_on) {
          const auto bt3 = std::chrono::steady_clock::now();
          bt_fill += std::chrono::duration<double, std::milli>(G.bt1 - G.bt0).count();
          bt_build += std::chrono::duration<double, std::milli>(G.bt2 - G.bt1).count();
          bt_iter += std::chrono::duration<double, std::milli>(bt3 - G.bt2).count();
          bt_groups += 1;
        }
        n_slots = 0;
        G.phase = Group::Fill;
        return 1;
      };
      // ONE group of the thread fills at a time (the others wait their turn): its pairs are complete and iterating while the next
      // group's are still being copied in, so the thread's groups run staggered -- copies under sweeps -- instead of in phase
      size_t fill_owner = 0;
      for (size_t retired = 0; retired < groups.size();) {
        bool moved = false;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
          Group& G = groups[gi];
          if (G.phase == Group::Retired) continue;
          if (G.phase == Group::Fill && gi != fill_owner) {
            if (groups[fill_owner].phase == Group::Fill) continue;
            fill_owner = gi;
          }
          const int r = group_step(G);
          if (r < 0) return;
          if (r == 2) ++retired;
          moved = moved || r != 0;
        }
#if defined(__x86_64__)
        if (!moved) __builtin_ia32_pause();
#endif
      }
      return;
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
    c->prof.gicp_device_solves += p.gicp_device_solves;
    c->prof.gicp_host_solves += p.gicp_host_solves;
    c->prof.gicp_quadratic_solves += p.gicp_quadratic_solves;
    c->prof.sources_adopted += p.sources_adopted;
    c->prof.grid_adopted += p.grid_adopted;
    if (p.brute_bound_worst > c->prof.brute_bound_worst) c->prof.brute_bound_worst = p.brute_bound_worst;
    std::memset(&p, 0, sizeof(p));
  }
  {
    static const bool trace = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_BATCH_TRACE"); return e && std::atoi(e) != 0; }();
    if (trace)
      fprintf(stderr, "[icpgpu] batch trace: call of %zu pairs took %.2f ms (+ %.2f ms creating worker contexts); device allocations so far in this process: %llu, %.2f ms of host time\n", n_pairs,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), create_ms, g_alloc_calls.load(), g_alloc_us.load() * 1e-3);
  }
  for (const ThreadError& e : errors)  // the first failure in thread order (each thread stops at its first)
    if (e.code != ICPGPU_OK) {
      c->err = "align_batch pair " + std::to_string(e.pair) + ": " + e.msg;
      return e.code;
    }
  return ICPGPU_OK;
}

}  // extern "C"
