// icp_multi.cpp -- icpgpu_align_batch_multi: independent scan pairs over the GPUs of one node from ONE process
// ("single process + ncclCommInitAll over the devices, one host thread per GPU", contiguous shards, one
// all-gather of fixed-size records at the end -- the C++ counterpart of icpslam_amd/sharding.py, for a host like the
// reference's, which is one C++ process: /root/reference/src/icpslam_node.cpp:3-14).
//
// No data-path collective: shard r = pairs [lo_r, hi_r) is solved by icpgpu_align_batch on device r's context, by its own
// host thread.  The gather: every shard's records (23 float64 = 184 B each, the layout of sharding.py) go to that device's
// send buffer, padded to the largest shard; the threads are joined, and ONLY IF every entry got that far the calling thread
// issues ONE all-gather over all devices (ncclGroupStart, one ncclAllGather per device, ncclGroupEnd; RCCL is loaded with
// dlopen: libicpgpu.so does not link it) -- a collective is entered by every rank or by none -- and waits for each stream
// with a deadline (ICPGPU_WAIT_TIMEOUT_MS).  All records are then on every device; entry 0's copy comes back to the host, is
// checked (every pair id exactly once, in order) and returned.  ICPGPU_COMM_HOST replaces RCCL's collective by a host-staged exchange through the same buffers
// (tests on a box with one GPU: several entries may name the same device then, which ncclCommInitAll refuses).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/icpgpu.h"

namespace icpgpu {
void set_ctx_host_share(icpgpu_ctx* ctx, int peers);  // icpgpu_batch.cpp: how many batch drivers share this process's CPUs with ctx
}

namespace {

constexpr int kRecordLen = 23;  // pair_id, iterations, converged, state, n_corr, mse, fitness, T[16] row-major (sharding.py)

constexpr size_t kAbiParams10 = 56, kAbiResult10 = 112;  // the shortest structs a caller may hand over (icp_ctx.h holds the same constants)

thread_local std::string g_multi_error;
int multi_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_multi_error = buf;
  return code;
}

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------------
typedef void* ncclComm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;  // optional
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl R = [] {
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return r;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
    return r;
  }();
  return R;
}
constexpr int kNcclFloat64 = 8;  // ncclDouble (rccl.h: ncclFloat64 = 8)

// ---- per-process state: one context per entry of the device list, communicators per device list -------------------------
struct Entry {
  icpgpu_ctx* ctx = nullptr;
  double *d_send = nullptr, *d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;  // doubles
  hipStream_t stream = nullptr;
};
struct State {
  std::mutex m;  // one multi call at a time
  std::map<std::pair<int, int>, Entry> entries;  // (device, ordinal among equal devices) -> context + buffers
  std::vector<int> comm_devices;
  std::vector<ncclComm_t> comms;
};
State& state() {
  static State* s = new State;  // leaked: nothing of HIP is torn down by a static destructor
  return *s;
}

int ensure_dev(double*& p, size_t& cap, size_t want) {
  if (cap >= want) return 0;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  if (hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(double)) != hipSuccess) return -1;
  cap = want;
  return 0;
}

void shard(size_t n, int r, int world, size_t& lo, size_t& hi) {  // sharding.shard_range: contiguous, sizes differ by <= 1
  const size_t base = n / (size_t)world, extra = n % (size_t)world;
  lo = (size_t)r * base + std::min<size_t>((size_t)r, extra);
  hi = lo + base + ((size_t)r < extra ? 1 : 0);
}

// The gather's deadline (ICPGPU_WAIT_TIMEOUT_MS, default 30 s -- the same switch as the solve's mailbox wait): a peer that
// never enters the collective (a dead device, a hung kernel on its stream) must become an error code, not a hang.
double gather_timeout_ms() {
  static const double v = [] {
    const char* e = std::getenv("ICPGPU_WAIT_TIMEOUT_MS");
    const double t = e ? std::atof(e) : 0.0;
    return t > 0.0 ? t : 30000.0;
  }();
  return v;
}
// 0: the stream drained; 1: it reported an error; 2: the deadline passed
int wait_stream(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return 1;
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (first ~ms: spin, a gather is microseconds)
    if ((spins & 63) == 63 &&
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > gather_timeout_ms())
      return 2;
  }
}

}  // namespace

extern "C" {

const char* icpgpu_multi_last_error(void) { return g_multi_error.c_str(); }

static int align_batch_multi_impl(const int* devices, int n_devices, const icpgpu_params* params, size_t n_pairs,
                                  const float* const* src, const size_t* n_src, const float* const* tgt, const size_t* n_tgt,
                                  int want_fitness, icpgpu_result* results, double* records, int communicator);

// the caller's structs may be shorter or longer than the library's (include/icpgpu.h, ABI rule): parameters are completed with the
// defaults, results are written with the caller's stride
int icpgpu_align_batch_multi_sz(const int* devices, int n_devices, const icpgpu_params* params, size_t n_pairs,
                                const float* const* src, const size_t* n_src, const float* const* tgt, const size_t* n_tgt,
                                int want_fitness, icpgpu_result* results, double* records, int communicator, size_t sizeof_params,
                                size_t sizeof_result) {
  g_multi_error.clear();
  if (sizeof_params < kAbiParams10 || sizeof_result < kAbiResult10 || sizeof_params > 4096 || sizeof_result > 4096)
    return multi_fail(ICPGPU_ERR_INVALID_ARG, "struct sizes %zu / %zu are not those of an icpgpu.h 1.x", sizeof_params, sizeof_result);
  icpgpu_params full;
  if (params) {
    icpgpu_default_params(&full);
    std::memcpy(&full, params, std::min(sizeof_params, sizeof(full)));
  }
  if (sizeof_result == sizeof(icpgpu_result) || !results)
    return align_batch_multi_impl(devices, n_devices, params ? &full : nullptr, n_pairs, src, n_src, tgt, n_tgt, want_fitness, results, records,
                                  communicator);
  std::vector<icpgpu_result> own(n_pairs);
  const int rc = align_batch_multi_impl(devices, n_devices, params ? &full : nullptr, n_pairs, src, n_src, tgt, n_tgt, want_fitness, own.data(),
                                        records, communicator);
  unsigned char* out = reinterpret_cast<unsigned char*>(results);
  for (size_t k = 0; k < n_pairs; ++k) {
    std::memset(out + k * sizeof_result, 0, sizeof_result);
    std::memcpy(out + k * sizeof_result, &own[k], std::min(sizeof_result, sizeof(icpgpu_result)));
  }
  return rc;
}

static int align_batch_multi_impl(const int* devices, int n_devices, const icpgpu_params* params, size_t n_pairs,
                                  const float* const* src, const size_t* n_src, const float* const* tgt, const size_t* n_tgt,
                                  int want_fitness, icpgpu_result* results, double* records, int communicator) {
  g_multi_error.clear();
  if (!devices || n_devices < 1 || n_devices > 64) return multi_fail(ICPGPU_ERR_INVALID_ARG, "bad device list");
  if (n_pairs && (!src || !n_src || !tgt || !n_tgt || !results)) return multi_fail(ICPGPU_ERR_INVALID_ARG, "null argument");
  if (communicator != ICPGPU_COMM_NONE && communicator != ICPGPU_COMM_RCCL && communicator != ICPGPU_COMM_HOST)
    return multi_fail(ICPGPU_ERR_INVALID_ARG, "bad communicator %d", communicator);
  if (communicator != ICPGPU_COMM_NONE && n_pairs && !records) return multi_fail(ICPGPU_ERR_INVALID_ARG, "records is null");
  State& S = state();
  std::lock_guard<std::mutex> guard(S.m);

  // contexts: entry r works on devices[r]; equal device numbers get distinct contexts
  std::vector<Entry*> E((size_t)n_devices);
  {
    std::map<int, int> seen;
    for (int r = 0; r < n_devices; ++r) {
      Entry& e = S.entries[{devices[r], seen[devices[r]]++}];
      if (!e.ctx) {
        const int rc = icpgpu_create(&e.ctx, devices[r]);
        if (rc != ICPGPU_OK) return multi_fail(rc, "device %d: %s", devices[r], icpgpu_last_error(nullptr));
        void* st = nullptr;
        icpgpu_get_stream(e.ctx, &st);
        e.stream = static_cast<hipStream_t>(st);
      }
      if (params) {
        const int rc = icpgpu_set_params(e.ctx, params);
        if (rc != ICPGPU_OK) return multi_fail(rc, "device %d: %s", devices[r], icpgpu_last_error(e.ctx));
      }
      E[(size_t)r] = &e;
    }
  }
  // communicator over exactly this device list (cached)
  const bool use_rccl = communicator == ICPGPU_COMM_RCCL;
  if (use_rccl) {
    Rccl& R = rccl();
    if (!R.ok) {
      const char* why = dlerror();                 // (reading it clears it: once)
      return multi_fail(ICPGPU_ERR_UNSUPPORTED, "librccl.so could not be loaded: %s", why ? why : "missing symbols");
    }
    const std::vector<int> want(devices, devices + n_devices);
    if (S.comm_devices != want) {
      for (ncclComm_t cm : S.comms) (void)R.CommDestroy(cm);
      S.comms.assign((size_t)n_devices, nullptr);
      S.comm_devices.clear();
      const int rc = R.CommInitAll(S.comms.data(), n_devices, devices);
      if (rc != 0) {
        S.comms.clear();
        return multi_fail(ICPGPU_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s (a device named twice? use ICPGPU_COMM_HOST)",
                          n_devices, R.GetErrorString(rc));
      }
      S.comm_devices = want;
    }
  }

  const size_t cap = (n_pairs + (size_t)n_devices - 1) / (size_t)n_devices;  // records per shard, padded
  const size_t block = cap * kRecordLen;                                      // doubles per shard
  std::vector<int> rcs((size_t)n_devices, ICPGPU_OK);
  std::vector<std::string> msgs((size_t)n_devices);
  std::vector<double> host_all(communicator == ICPGPU_COMM_NONE ? 0 : block * (size_t)n_devices, -1.0);
  std::vector<double> gathered(communicator == ICPGPU_COMM_NONE ? 0 : block * (size_t)n_devices, -2.0);
  auto failed = [&](int r, int code, const std::string& why) {
    if (rcs[(size_t)r] == ICPGPU_OK) {
      rcs[(size_t)r] = code;
      msgs[(size_t)r] = why;
    }
  };
  auto first_failure = [&]() -> int {
    for (int r = 0; r < n_devices; ++r)
      if (rcs[(size_t)r] != ICPGPU_OK) return multi_fail(rcs[(size_t)r], "%s", msgs[(size_t)r].c_str());
    return ICPGPU_OK;
  };

  // ---- phase 1, one host thread per entry: solve the shard, pack its records, put them into the entry's send buffer.
  // Nothing here waits for another entry.
  auto work = [&](int r) {
    Entry& e = *E[(size_t)r];
    size_t lo, hi;
    shard(n_pairs, r, n_devices, lo, hi);
    icpgpu::set_ctx_host_share(e.ctx, n_devices);  // this context's batch shares the process's CPUs with n_devices - 1 others
    if (hi > lo) {
      const int rc = icpgpu_align_batch(e.ctx, hi - lo, src + lo, n_src + lo, tgt + lo, n_tgt + lo, want_fitness, results + lo);
      if (rc != ICPGPU_OK) failed(r, rc, std::string("shard ") + std::to_string(r) + ": " + icpgpu_last_error(e.ctx));
    }
    icpgpu::set_ctx_host_share(e.ctx, 1);
    if (communicator == ICPGPU_COMM_NONE || block == 0 || rcs[(size_t)r] != ICPGPU_OK) return;
    std::vector<double> mine(block, -1.0);
    for (size_t k = lo; k < hi; ++k) {
      double* rec = mine.data() + (k - lo) * kRecordLen;
      const icpgpu_result& R = results[k];
      rec[0] = (double)k;
      rec[1] = R.iterations;
      rec[2] = R.converged ? 1.0 : 0.0;
      rec[3] = R.convergence_state;
      rec[4] = R.n_correspondences;
      rec[5] = R.mse_last;
      rec[6] = R.fitness;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) rec[7 + 4 * i + j] = (double)R.T[j * 4 + i];  // column-major result -> row-major record
    }
    if (hipSetDevice(devices[r]) != hipSuccess || ensure_dev(e.d_send, e.send_cap, block) != 0 ||
        ensure_dev(e.d_recv, e.recv_cap, block * (size_t)n_devices) != 0)
      return failed(r, ICPGPU_ERR_HIP, "gather buffers on device " + std::to_string(devices[r]));
    if (hipMemcpyAsync(e.d_send, mine.data(), block * sizeof(double), hipMemcpyHostToDevice, e.stream) != hipSuccess ||
        wait_stream(e.stream) != 0)   // (`mine` is pageable memory local to this thread: it must be consumed before it goes)
      failed(r, ICPGPU_ERR_HIP, "upload of the records on device " + std::to_string(devices[r]));
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < n_devices; ++r) threads.emplace_back(work, r);
  work(0);
  for (auto& t : threads) t.join();
  // ---- agreement BEFORE the collective: the join above is the barrier, rcs[] the shared verdict.  If any entry failed --
  // its shard, its buffers, its upload -- NO entry enters the gather and the call returns that entry's error: a collective
  // is entered by every rank or by none (until round 4 a rank without buffers skipped ncclAllGather while the others waited in it).
  if (int rc = first_failure()) return rc;
  if (communicator == ICPGPU_COMM_NONE || n_pairs == 0) return ICPGPU_OK;

  // ---- phase 2, this thread: ONE collective over all entries -- the all-gathers of every device issued between
  // ncclGroupStart / ncclGroupEnd (RCCL's single-thread-many-devices form), then every stream waited for with a deadline.
  if (use_rccl) {
    Rccl& R = rccl();
    int rc = R.GroupStart();
    for (int r = 0; r < n_devices && rc == 0; ++r)
      rc = R.AllGather(E[(size_t)r]->d_send, E[(size_t)r]->d_recv, block, kNcclFloat64, S.comms[(size_t)r], E[(size_t)r]->stream);
    const int rc_end = R.GroupEnd();   // (always closed, so that a failed enqueue does not leave the group open)
    if (rc == 0) rc = rc_end;
    bool timed_out = false;
    for (int r = 0; r < n_devices && rc == 0; ++r) {
      const int w = wait_stream(E[(size_t)r]->stream);
      if (w != 0) {
        timed_out = w == 2;
        failed(r, ICPGPU_ERR_HIP, timed_out ? "ncclAllGather did not complete within " + std::to_string((long long)gather_timeout_ms()) +
                                                  " ms on device " + std::to_string(devices[r])
                                            : "stream failure after ncclAllGather on device " + std::to_string(devices[r]));
        break;
      }
    }
    if (rc != 0) failed(0, ICPGPU_ERR_HIP, std::string("ncclAllGather: ") + R.GetErrorString(rc));
    if (rc != 0 || timed_out || first_failure() != ICPGPU_OK) {
      // the communicators are in an unknown state: drop them (abort if the library offers it, destroying a communicator with
      // a collective in flight may block) and let the next call build new ones
      for (ncclComm_t cm : S.comms)
        if (cm) (void)(R.CommAbort ? R.CommAbort(cm) : R.CommDestroy(cm));
      S.comms.clear();
      S.comm_devices.clear();
      return first_failure();
    }
  } else {
    // host-staged exchange through the same device buffers: every block down, then everybody's blocks up
    for (int r = 0; r < n_devices; ++r) {
      Entry& e = *E[(size_t)r];
      if (hipMemcpyAsync(host_all.data() + (size_t)r * block, e.d_send, block * sizeof(double), hipMemcpyDeviceToHost, e.stream) != hipSuccess ||
          wait_stream(e.stream) != 0)
        failed(r, ICPGPU_ERR_HIP, "host-staged gather (download) on device " + std::to_string(devices[r]));
    }
    if (int rc = first_failure()) return rc;
    for (int r = 0; r < n_devices; ++r) {
      Entry& e = *E[(size_t)r];
      if (hipMemcpyAsync(e.d_recv, host_all.data(), block * (size_t)n_devices * sizeof(double), hipMemcpyHostToDevice, e.stream) != hipSuccess ||
          wait_stream(e.stream) != 0)
        failed(r, ICPGPU_ERR_HIP, "host-staged gather (upload) on device " + std::to_string(devices[r]));
    }
    if (int rc = first_failure()) return rc;
  }
  if (hipSetDevice(devices[0]) != hipSuccess ||
      hipMemcpy(gathered.data(), E[0]->d_recv, block * (size_t)n_devices * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return multi_fail(ICPGPU_ERR_HIP, "read-back of the gathered records");
  // unpack: shard r's records sit at r * block; together they must name every pair exactly once, in order
  size_t k = 0;
  for (int r = 0; r < n_devices; ++r) {
    size_t lo, hi;
    shard(n_pairs, r, n_devices, lo, hi);
    for (size_t i = lo; i < hi; ++i, ++k) {
      const double* rec = gathered.data() + (size_t)r * block + (i - lo) * kRecordLen;
      if (rec[0] != (double)i) return multi_fail(ICPGPU_ERR_HIP, "gathered records do not cover the pair ids (slot %zu holds %g)", i, rec[0]);
      std::memcpy(records + i * kRecordLen, rec, kRecordLen * sizeof(double));
    }
  }
  return k == n_pairs ? ICPGPU_OK : multi_fail(ICPGPU_ERR_HIP, "gather size mismatch");
}

}  // extern "C"
