// icpgpu_p2p.cpp -- point-to-point ICP behind icpgpu_align (pcl::IterativeClosestPoint semantics; rows a2-a10).
// Per iteration: search kernel with the fused reduction (a2+a3+a4, a6 applied on load) -> reduce_final_kernel stores the 17 sums
// as self-tagged 16-byte pairs into the pinned host mailbox -> the host polls the tags -> Umeyama / SVD (a5) + convergence test
// (a7) in float64 -> next launch.  No copy engine and no stream synchronisation on the iteration path.
#include "icp_ctx.h"


namespace icpgpu_impl {

int nn_keys_brute(icpgpu_ctx* c, const float4* tgt_pts, int n_t, const Xform& T, unsigned long long* keys, bool* used_mfma) {
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
        const char* chk = ICPGPU_DEV_ENV("ICPGPU_MFMA_CHECK_BOUND");
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
int resolve_sweep_timings(icpgpu_ctx* c, bool block) {
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
double wait_timeout_ms() {
  static const double v = [] { const char* e = std::getenv("ICPGPU_WAIT_TIMEOUT_MS"); const double x = e ? std::atof(e) : 0.0; return x > 0.0 ? x : 30000.0; }();
  return v;
}

// the sums mailbox: pair k = {sum bits, tag(sweep number, bits)} at words 2k, 2k + 1 (reduce_final_kernel; icp_kernels.h:
// mailbox_tag): a pair counts when its tag carries the number AND the checksum of the bits beside it
bool flags_ready(const volatile unsigned long long* pairs, int n_pairs, unsigned long long seq) {
  for (int k = 0; k < n_pairs; ++k)
    if ((pairs[2 * k + 1] >> 24) != seq) return false;  // the numbers first (cheap), the checksums once they are all there
  unsigned long long bits;
  for (int k = 0; k < n_pairs; ++k)
    if (!mailbox_read(pairs + 2 * k, seq, &bits)) return false;
  return true;
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



int sweep_issue(icpgpu_ctx* c, const Xform& T, float thr, bool open_range, SweepTicket& tk) {
  static const bool timing = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_P2P_TIMING"); return e && std::atoi(e) != 0; }();
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
  if (timed && !ev[0])  // the ring's events are created when a slot is first timed (192 events per context up front were a third of
    for (int k = 0; k < 3; ++k) HIP_TRY(c, hipEventCreate(&ev[k]));  // icpgpu_create's millisecond, and a batch creates 64 contexts)
#define EVREC(e) do { if (timed) HIP_TRY(c, hipEventRecord((e), c->stream)); } while (0)
  EVREC(ev[0]);
  const float4* red_src = nullptr;
  int red_n = 0;
  volatile int* few_host = nullptr;
#if defined(ICPGPU_DEV_SWITCHES)
  // EXPERIMENT, development flavour only (libicpgpu_dev.so; icp_tile.hip is not part of the release library): the grid search
  // on the matrix cores -- bit-identical results, faster at 50k x 50k, slower at 200k x 200k (experiments).
  // Read per sweep so that a test can switch it.
  const char* tile_env = ICPGPU_DEV_ENV("ICPGPU_TILE_SEARCH");
  const int tile_search = tile_env ? std::atoi(tile_env) : 0;
#endif
  if (false) {
#if defined(ICPGPU_DEV_SWITCHES)
  } else if (use_grid && !open_range && tile_search && source_ordered(c) && n_s >= kMfmaMinPoints) {
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
    HIP_TRY(c, launch_reduce(red_src, red_n, c->tgt.data(), keys, T, thr, partials, d_sums, c->h_flags_dev, wire_seq(c, seq), c->stream));
#endif
  } else if (use_grid && !open_range) {
    // cell-ordered source when there is one (non-finite points are absent from it: they never match anyway)
    const bool ordered = source_ordered(c);
    const float4* src_pts = ordered ? static_cast<const float4*>(c->src_grid.sorted.ptr) : c->src.data();
    const int n_q = ordered ? c->src_grid.n_binned : n_s;
    const int blocks = grid_search_blocks(n_q);
    if ((rc = ensure(c, c->partials, (size_t)blocks * kReduceTerms * sizeof(double)))) return rc;
    partials = static_cast<double*>(c->partials.ptr);
    unsigned int* prev = nullptr;
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
    HIP_TRY(c, launch_reduce_final(partials, blocks, /*term_major=*/true, d_sums, c->h_flags_dev, wire_seq(c, seq), c->stream));
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
    HIP_TRY(c, launch_reduce(red_src, red_n, c->tgt.data(), keys, T, thr, partials, d_sums, c->h_flags_dev, wire_seq(c, seq), c->stream));
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
                           c->h_sums_dev, c->h_flags_dev, wire_seq(c, seq2), c->stream));
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


// development flavour, ICPGPU_STAGE_DIRECT=0: result clouds go through device memory and the copy engine, as until round 5
static bool stage_direct() {
  static const bool v = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_STAGE_DIRECT"); return !e || std::atoi(e) != 0; }();
  return v;
}

int output_cloud_issue(icpgpu_ctx* c, const Xform& T, float* out_xyzw, StageTicket& tk) {
  tk.issued = false;
  const int n_s = (int)c->src.n;
  const size_t bytes = (size_t)n_s * sizeof(float4);
  if ((!out_xyzw && !c->want_view) || n_s == 0 || !stage_direct()) return ICPGPU_OK;
  if (bytes > kStageMaxBytes && !c->want_view) return ICPGPU_OK;
  int rc = ensure_stage(c, bytes, c->want_view);
  if (rc) return rc;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, launch_transform(c->src.data(), n_s, T, static_cast<float4*>(c->h_stage_dev), c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  return stage_post(c, nullptr, 0, tk);
}

int output_cloud_complete(icpgpu_ctx* c, StageTicket& tk, float* out_xyzw) {
  if (!tk.issued) return ICPGPU_OK;
  int rc = stage_wait(c, tk, nullptr);
  if (rc) return rc;
  const size_t n_s = c->src.n;
  if (out_xyzw && !c->want_view) std::memcpy(out_xyzw, c->h_stage, n_s * sizeof(float4));
  float ms = 0.f;
  hipError_t te = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
  if (te == hipErrorNotReady) {  // (the events lie in front of the marker's kernel: complete; the runtime may not have noticed yet)
    (void)hipGetLastError();
    HIP_TRY(c, hipEventSynchronize(c->ev[1]));
    te = hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
  }
  HIP_TRY(c, te);
  c->prof.transform_launches += 1;
  c->prof.transform_ms += ms;
  c->prof.transform_bytes += 32ull * (uint64_t)n_s;
  return ICPGPU_OK;
}

int write_output_cloud(icpgpu_ctx* c, const Xform& T, float* out_xyzw) {
  const int n_s = (int)c->src.n;
  if ((!out_xyzw && !c->want_view) || n_s == 0) return ICPGPU_OK;
  StageTicket tk;
  int rc = output_cloud_issue(c, T, out_xyzw, tk);
  if (rc) return rc;
  if (tk.issued) return output_cloud_complete(c, tk, out_xyzw);
  // too large for the staging buffer (or the development switch): device memory, then the copy engine
  if ((rc = ensure(c, c->out, (size_t)n_s * sizeof(float4)))) return rc;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, launch_transform(c->src.data(), n_s, T, static_cast<float4*>(c->out.ptr), c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  if (c->want_view) {
    if ((rc = ensure_stage(c, (size_t)n_s * sizeof(float4), true))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_stage, c->out.ptr, (size_t)n_s * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else if ((rc = copy_to_host(c, out_xyzw, c->out.ptr, (size_t)n_s * sizeof(float4)))) return rc;
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  c->prof.transform_launches += 1;
  c->prof.transform_ms += ms;
  c->prof.transform_bytes += 32ull * (uint64_t)n_s;
  return ICPGPU_OK;
}

void init_result(icpgpu_result* r) {
  std::memset(r, 0, sizeof(*r));
  for (int i = 0; i < 16; ++i) r->T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  r->fitness = NAN;
}



int p2p_finish(icpgpu_ctx* c, P2PRun& r) {
  const Xform Tf = to_xform(r.final_T);
  // (the aligned cloud was queued in front of the fitness sweep when there was one: p2p_advance)
  int rc = r.out_done ? ICPGPU_OK
                      : (r.out_ticket.issued ? output_cloud_complete(c, r.out_ticket, r.out_xyzw) : write_output_cloud(c, Tf, r.out_xyzw));
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
int p2p_advance(icpgpu_ctx* c, P2PRun& r, int* deferred) {
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
    // the aligned cloud first (into the pinned staging buffer, a marker behind it), then the fitness sweep: p2p_finish copies the
    // cloud out while nothing else waits for it (round 6; icp_odometer.cpp:196-201 asks for both on every scan)
    if ((rc = output_cloud_issue(c, to_xform(r.final_T), r.out_xyzw, r.out_ticket))) return rc;
    return sweep_issue(c, to_xform(r.final_T), FLT_MAX, true, r.ticket);
  }
  return p2p_finish(c, r);
}

int align_p2p(icpgpu_ctx* c, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* res) {
  P2PRun r;
  int rc = p2p_begin(c, r, guess, out_xyzw, want_fitness, res);
  static const bool timing = [] { const char* e = ICPGPU_DEV_ENV("ICPGPU_P2P_TIMING"); return e && std::atoi(e) != 0; }();
  while (!rc && r.phase != P2PRun::Done) {
    const auto t0 = std::chrono::steady_clock::now();
    if (r.phase == P2PRun::Fitness && r.out_ticket.issued) {  // the aligned cloud leaves the staging buffer while the fitness sweep runs
      if ((rc = output_cloud_complete(c, r.out_ticket, r.out_xyzw))) break;
      r.out_done = true;
    }
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

}  // namespace icpgpu_impl

extern "C" {

int icpgpu_align(icpgpu_ctx* c, const float* guess, float* out_xyzw, int want_fitness, icpgpu_result* user_res) {
  ENTER(c);
  if (!user_res) return fail(c, ICPGPU_ERR_INVALID_ARG, "result is null");
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "align: source and target must be set first");
  // a caller whose icpgpu_result is not this library's (include/icpgpu.h, ABI rule) gets the leading bytes it knows
  icpgpu_result own;
  icpgpu_result* res = c->abi_result == sizeof(icpgpu_result) ? user_res : &own;
  const int rc = c->params.method == ICPGPU_GICP ? align_gicp(c, guess, out_xyzw, want_fitness, res) : align_p2p(c, guess, out_xyzw, want_fitness, res);
  if (res != user_res) {
    std::memset(user_res, 0, c->abi_result);
    std::memcpy(user_res, res, std::min(c->abi_result, sizeof(own)));
  }
  return rc;
}

// ICPGPU_GICP_DEVICE=auto: time both inner solvers of GICP on the clouds the context holds (align_gicp alternates and times while
// gicp_choice is 0) and keep the faster; see include/icpgpu.h
int icpgpu_calibrate(icpgpu_ctx* c, int* choice) {
  ENTER(c);
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "calibrate: source and target must be set first");
  if (c->params.method != ICPGPU_GICP) return fail(c, ICPGPU_ERR_INVALID_ARG, "calibrate: the context's method is not GICP");
  if (gicp_device_solver_mode() == 2 && !gicp_inner_quadratic(c)) {
    const int before = c->gicp_choice;
    c->gicp_choice = 0;
    c->gicp_cal_us[0] = c->gicp_cal_us[1] = 0.0;
    c->gicp_cal_evals[0] = c->gicp_cal_evals[1] = 0;
    c->gicp_cal_runs[0] = c->gicp_cal_runs[1] = 0;
    icpgpu_result scratch;
    int rc = ICPGPU_OK;
    for (int k = 0; k < 12 && c->gicp_choice == 0 && rc == ICPGPU_OK; ++k) rc = align_gicp(c, nullptr, nullptr, 0, &scratch);
    if (c->gicp_choice == 0) c->gicp_choice = before;  // too few evaluations to tell (tiny clouds, runs the device solver cannot take)
    if (rc != ICPGPU_OK) return rc;
  }
  if (choice) *choice = c->gicp_choice == 2 && c->gicp_device_ok ? ICPGPU_GICP_SOLVER_DEVICE : ICPGPU_GICP_SOLVER_HOST;
  return ICPGPU_OK;
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

int icpgpu_align_view(icpgpu_ctx* c, const float* guess, int want_fitness, icpgpu_result* user_res, const float** view_xyzw, size_t* n_out) {
  ENTER(c);
  if (!user_res || !view_xyzw || !n_out) return fail(c, ICPGPU_ERR_INVALID_ARG, "null argument");
  *view_xyzw = nullptr;
  *n_out = 0;
  if (!c->src.set || !c->tgt.set) return fail(c, ICPGPU_ERR_NO_INPUT, "align: source and target must be set first");
  icpgpu_result own;
  icpgpu_result* res = c->abi_result == sizeof(icpgpu_result) ? user_res : &own;
  struct ViewGuard {
    icpgpu_ctx* c;
    ~ViewGuard() { c->want_view = false; }
  } guard{c};
  c->want_view = true;
  const int rc = c->params.method == ICPGPU_GICP ? align_gicp(c, guess, nullptr, want_fitness, res) : align_p2p(c, guess, nullptr, want_fitness, res);
  if (res != user_res) {
    std::memset(user_res, 0, c->abi_result);
    std::memcpy(user_res, res, std::min(c->abi_result, sizeof(own)));
  }
  if (rc == ICPGPU_OK && c->src.n) {
    *view_xyzw = static_cast<const float*>(c->h_stage);
    *n_out = c->src.n;
  }
  return rc;
}

int icpgpu_transform(icpgpu_ctx* c, const float* T, float* out_xyzw) {
  ENTER(c);
  if (!c->src.set) return fail(c, ICPGPU_ERR_NO_INPUT, "transform: no source set");
  if (c->src.n && !out_xyzw) return fail(c, ICPGPU_ERR_INVALID_ARG, "out is null");
  return write_output_cloud(c, to_xform(T), out_xyzw);
}

}  // extern "C"
