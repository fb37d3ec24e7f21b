// sc_multi.hip — the native multi-device entry of include/saccot.h: sc_create_multi / sc_register_multi.
//
// a C++ (or mex) host has no torch.distributed; it hands the library a list of devices and gets
// (R, t, mask) back.  One context (sc_ctx) per device, driven through the same phase API a one-process-per-GPU host
// uses (sc_shard_*_device) — this file is a CLIENT of the C ABI, it contains no kernel — and the four
// collectives of a call are RCCL's, on the devices' streams:
//     all-gather of the bit rows  |  all-reduce SUM of the 1 KiB sample histogram  |  all-gather of the candidate
//     blobs  |  all-gather of the 16-byte winner key pairs
// Build decisions, explicit as SURVEY §8(b) asks:
//  * one WORKER THREAD per device instead of one host thread for all: a call is ~35 kernel launches and four host
//    read-backs per device, i.e. ~1 ms of launch time per step for eight devices from a single thread — more than the
//    GPUs' own time on the 8-GPU configs.  The workers live as long as the sc_multi; the caller's thread only wakes
//    them and waits.  Each worker issues the collectives on its own communicator (RCCL's multi-thread form; no
//    ncclGroupStart/End needed).  The calling convention stays single-caller, like sc_ctx.
//  * RCCL is opened at run time (dlopen "librccl.so.1") when n_dev > 1: n_dev == 1 makes no RCCL call at all and
//    the library carries no link-time dependency on it.
//  * every wait between threads SPINS (with a pause, then yields): the workers are dedicated threads and a step is a few
//    hundred microseconds, so a mutex + condition-variable wake-up (5-50 us each, ten per call in round 2) would be a
//    large part of it.  Idle workers fall asleep on a condition variable after ~200 us without a job.
//  * inputs go through ONE pinned, device-mapped host area every device reads directly (the staging kernel streams the
//    24 n bytes over the host link), outputs come back through another one rank 0's finalize kernel writes: no
//    pageable-memory hipMemcpy on the path (10-20 us of driver time each), exactly like sc_register.
//  * errors: nobody enters a collective unless every rank finished the phase before it (agreement BEFORE: a rank must
//    not compute on another rank's garbage), and every rank learns right AFTER the enqueue whether it succeeded everywhere
//    (a collective one rank never joined blocks the others' streams for good: the communicators are then aborted and the
//    handle refuses further calls).  Both agreements are spin barriers.
//  * a LOOPBACK transport (sc_create_multi_loopback: n ranks on ONE device, device copies between the ranks' buffers
//    behind host barriers) exists so that the whole orchestration — buffers, phase order, error agreement, outputs —
//    is exercised bit for bit on a one-GPU box.  Only the RCCL calls themselves are then untested.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: every function is resolved with dlsym

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/saccot.h"
#include "../../include/saccot_debug.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool open(std::string& err) {
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { err = std::string("cannot open librccl.so.1: ") + dlerror(); return false; }
#define SC_SYM(field, name) \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); \
    if (!field) { err = std::string("librccl lacks ") + name; return false; }
    SC_SYM(CommInitAll, "ncclCommInitAll") SC_SYM(CommDestroy, "ncclCommDestroy") SC_SYM(AllGather, "ncclAllGather")
    SC_SYM(AllReduce, "ncclAllReduce") SC_SYM(GetErrorString, "ncclGetErrorString")
#undef SC_SYM
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
    return true;
  }
};

inline void spin_pause() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}
// spin on `done()`; after a while give the core away between polls (more ranks than cores: the loopback tests)
template <class F> inline void spin_until(F&& done) {
  for (uint32_t spins = 0; !done(); spins++) {
    if (spins < 4096) spin_pause();
    else std::this_thread::yield();
  }
}

// Sense-reversing SPIN barrier of the worker threads; also agrees on the first non-OK status reported this call.
// Every worker arrives at every barrier of a call, whatever happened to it (it reports its status and skips the work).
struct Barrier {
  int n = 0;
  std::atomic<uint32_t> count{0}, sense{0};
  std::atomic<int> worst{SC_OK};
  int arrive(int status) {  // returns the worst status any worker has reported so far this call
    if (status != SC_OK) { int expected = SC_OK; worst.compare_exchange_strong(expected, status); }
    const uint32_t s = sense.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)n) {
      count.store(0, std::memory_order_relaxed);        // before the flip that releases the others
      sense.store(s + 1, std::memory_order_release);
    } else {
      spin_until([&] { return sense.load(std::memory_order_acquire) != s; });
    }
    return worst.load(std::memory_order_acquire);
  }
};

struct Rank {
  int device = 0;
  sc_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  // exchange buffers (device memory of this rank's device), grown on demand
  void *bits = nullptr, *cand = nullptr, *src = nullptr, *tgt = nullptr, *mask = nullptr;
  size_t bits_cap = 0, cand_cap = 0, src_cap = 0, tgt_cap = 0, mask_cap = 0;
  uint32_t* hist = nullptr;
  uint64_t* keys = nullptr;
  float* Rt = nullptr;
  std::thread worker;
  int status = SC_OK;
  sc_stats stats{};
  std::string error;
};

}  // namespace

struct sc_multi {
  int n = 0;
  bool loopback = false;
  bool workers = false;  // the rank machinery (phase API + collectives); false: n == 1, plain sc_register
  std::atomic<bool> broken{false};   // a collective failed on some rank and the communicators were aborted: no further calls
  std::vector<Rank> ranks;
  Rccl rccl;
  Barrier bar;
  std::string last_error;
  // job hand-off to the workers: they spin on `job` for a while after a call, then sleep on cv_go
  std::atomic<uint64_t> job{0};
  std::atomic<int> running{0}, sleepers{0};
  std::atomic<bool> quit{false};
  std::mutex m;
  std::condition_variable cv_go;
  // the call in flight
  const float *src = nullptr, *tgt = nullptr;
  int64_t npts = 0;
  sc_params params{};
  int cand_level = 0;  // sticky: raised whenever a call came back with SC_ERETRY (candidate blobs too small)
  bool estimate = true;  // SC_FLAG_EST_BOUND (stage B pruned by an estimated bound, no histogram all-reduce); sticky off once estimates have FAILED twice
  bool certify_once = false;  // the running call is a repeat after SC_EBOUND: without the flag, this once
  int est_fails = 0;
  float* R = nullptr; float* t = nullptr; uint8_t* mask = nullptr;
  // pinned, device-mapped staging (portable: every device reads h_in, rank 0's finalize kernel writes h_out)
  void* h_in = nullptr; size_t h_in_cap = 0;
  void* h_out = nullptr; size_t h_out_cap = 0;
  bool pinned_io = false;  // this call's inputs / outputs go through h_in / h_out
};

namespace {

constexpr int64_t PINNED_MAX_N = 1 << 20;  // as sc_register: beyond 1 M correspondences plain copies

#define MHIP(rk, expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) { (rk).error = std::string(#expr) + ": " + hipGetErrorString(_e); return SC_EHIP; } \
  } while (0)

int grow(Rank& rk, void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return SC_OK;
  if (*p) { MHIP(rk, hipStreamSynchronize(rk.stream)); MHIP(rk, hipFree(*p)); *p = nullptr; *cap = 0; }
  if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); rk.error = "hipMalloc failed (exchange buffer)"; return SC_ENOMEM; }
  *cap = bytes;
  return SC_OK;
}

// ---- RCCL transport: in-place all-gather of `per` bytes per rank / SUM all-reduce of the histogram, on the rank's stream
int rccl_allgather(sc_multi* M, int r, void* buf, size_t per) {
  Rank& rk = M->ranks[r];
  char* mine = static_cast<char*>(buf);
  const ncclResult_t e = M->rccl.AllGather(mine + (size_t)r * per, mine, per, ncclUint8, rk.comm, rk.stream);
  if (e != ncclSuccess) { rk.error = std::string("ncclAllGather: ") + M->rccl.GetErrorString(e); return SC_ERCCL; }
  return SC_OK;
}
int rccl_allreduce_hist(sc_multi* M, int r) {
  Rank& rk = M->ranks[r];
  const ncclResult_t e = M->rccl.AllReduce(rk.hist, rk.hist, SC_HIST_WORDS, ncclUint32, ncclSum, rk.comm, rk.stream);
  if (e != ncclSuccess) { rk.error = std::string("ncclAllReduce: ") + M->rccl.GetErrorString(e); return SC_ERCCL; }
  return SC_OK;
}

// ---- loopback transport (all ranks on one device): device copies between the ranks' buffers behind the barrier.
// `ok`: this rank and, as far as it knows, everybody else is fine — a rank that is not still ARRIVES at both inner
// barriers (skipping the copies): a missing arrival would deadlock every worker.  Returns this rank's own status.
template <class Buf> int loop_allgather(sc_multi* M, int r, bool ok, Buf&& buf_of, size_t per) {
  Rank& rk = M->ranks[r];
  auto step = [&](auto&& body) -> int { return ok ? body() : SC_OK; };
  // every rank's slice must exist before anyone copies it ...
  int e = step([&]() -> int { MHIP(rk, hipStreamSynchronize(rk.stream)); return SC_OK; });
  if (M->bar.arrive(e) != SC_OK) ok = false;
  int e2 = step([&]() -> int {
    char* mine = static_cast<char*>(buf_of(rk));
    for (int q = 0; q < M->n; q++)
      if (q != r)
        MHIP(rk, hipMemcpyAsync(mine + (size_t)q * per, static_cast<char*>(buf_of(M->ranks[q])) + (size_t)q * per, per,
                                hipMemcpyDeviceToDevice, rk.stream));
    MHIP(rk, hipStreamSynchronize(rk.stream));
    return SC_OK;
  });
  M->bar.arrive(e2);  // ... and nobody may move on (and overwrite its slice) before everyone has read it
  return e != SC_OK ? e : e2;
}
int loop_allreduce_hist(sc_multi* M, int r, bool ok) {
  Rank& rk = M->ranks[r];
  auto step = [&](auto&& body) -> int { return ok ? body() : SC_OK; };
  int e = step([&]() -> int { MHIP(rk, hipStreamSynchronize(rk.stream)); return SC_OK; });
  if (M->bar.arrive(e) != SC_OK) ok = false;
  std::vector<uint32_t> sum(SC_HIST_WORDS, 0u), one(SC_HIST_WORDS);
  int e2 = step([&]() -> int {
    for (int q = 0; q < M->n; q++) {
      MHIP(rk, hipMemcpy(one.data(), M->ranks[q].hist, SC_HIST_WORDS * 4, hipMemcpyDeviceToHost));
      for (int b = 0; b < SC_HIST_WORDS; b++) sum[b] += one[b];
    }
    return SC_OK;
  });
  if (M->bar.arrive(e2) != SC_OK) ok = false;  // everyone has read every histogram before anyone overwrites its own
  int e3 = step([&]() -> int { MHIP(rk, hipMemcpy(rk.hist, sum.data(), SC_HIST_WORDS * 4, hipMemcpyHostToDevice)); return SC_OK; });
  return e != SC_OK ? e : (e2 != SC_OK ? e2 : e3);
}

// RCCL path, after a collective failed to enqueue on some rank: the others' streams wait for a peer that never joins.
// Abort the communicators (that ends the stuck kernels) — the handle is unusable afterwards.
void abort_comms(sc_multi* M, int r) {
  Rank& rk = M->ranks[r];
  if (rk.comm && M->rccl.CommAbort) { (void)M->rccl.CommAbort(rk.comm); rk.comm = nullptr; }
  M->broken.store(true);
}

// One rank's part of a call.  Every rank reaches the same barriers in the same order whatever happens: a rank that
// has failed keeps arriving (with its status) and skips the work; nobody enters a collective unless every rank
// got there without an error, and every rank knows right after a collective's enqueue whether all enqueues succeeded.
int run_rank(sc_multi* M, int r) {
  Rank& rk = M->ranks[r];
  const int G = M->n;
  const int64_t n = M->npts;
  sc_params p = M->params;
  p.shard_rank = r; p.shard_world = G;
  p.shard_cand_level = M->cand_level;
  // (only on graphs below 8192 correspondences: beyond that a replicated sample costs more than the shared certifying one — measured
  // by tools/emulate_world.py at C3, see sac-cot_amd/shard.py)
  const bool est = M->estimate && !M->certify_once && G > 1 && n < 8192;
  // r04b: on those small graphs stages A and B are REPLICATED — every device runs them for the whole job, pruned by the estimated bound
  // (sc_hypothesize_device with SC_FLAG_EST_BOUND), and scores its share: ONE exchange of the 16-byte key pairs per call instead of
  // three collectives (one rank's emulated step, C2 weak / C4 strong at 8 ranks: 0.26 / 0.27 ms against 0.31 / 0.32 sharded).
  // SC_FLAG_SHARD_AB keeps the sharded form; so does a context whose estimate failed once (M->estimate).
  const bool replicated = est && !(p.flags & SC_FLAG_SHARD_AB);
  p.flags &= ~SC_FLAG_SHARD_AB;
  if (est) p.flags |= SC_FLAG_EST_BOUND; else p.flags &= ~SC_FLAG_EST_BOUND;
  if (p.shard_block == 0) p.shard_block = 1024;
  int rc = SC_OK;
  bool aborted = false;
  sc_shard_plan plan; plan.size = sizeof plan;
  auto fail_from_ctx = [&](int code) { rk.error = sc_last_error(rk.ctx); return code; };
  auto compute = [&](auto&& body) {  // a phase of this rank's own work, then agreement BEFORE the collective that follows
    if (rc == SC_OK && !aborted) rc = body();
    const int worst = M->bar.arrive(rc);
    if (rc == SC_OK && worst != SC_OK) rc = worst;  // another rank failed: stop here too (status of the first failure)
  };
  auto collective = [&](auto&& rccl_body, auto&& loop_body) {
    if (M->loopback) {  // the transport's own barriers: every rank arrives, failed or not
      const int e = loop_body(rc == SC_OK && !aborted);
      if (rc == SC_OK) rc = e;
      const int worst = M->bar.worst.load(std::memory_order_acquire);
      if (rc == SC_OK && worst != SC_OK) rc = worst;
      return;
    }
    int e = SC_OK;
    if (rc == SC_OK && !aborted) e = rccl_body();
    const int worst = M->bar.arrive(e);  // agreement AFTER the enqueue
    if (rc == SC_OK && e != SC_OK) rc = e;
    if (worst == SC_ERCCL && !aborted) { abort_comms(M, r); aborted = true; if (rc == SC_OK) rc = SC_ERCCL; }
    else if (rc == SC_OK && worst != SC_OK) rc = worst;
  };
  const float *d_src = nullptr, *d_tgt = nullptr;
  float* d_Rt = nullptr; uint8_t* d_mask = nullptr;
  auto setup_io = [&](bool sharded) -> int {  // this rank's buffers and where its inputs / outputs live
    MHIP(rk, hipSetDevice(rk.device));
    int e = SC_OK;
    if (sharded) {
      e = sc_shard_plan_query(&p, n, &plan);
      if (e) { rk.error = "bad parameters"; return e; }
      if ((e = grow(rk, &rk.bits, &rk.bits_cap, plan.bits_bytes_total))) return e;
      if ((e = grow(rk, &rk.cand, &rk.cand_cap, (size_t)G * plan.cand_bytes_per_rank))) return e;
    }
    if (M->pinned_io) {  // every device reads the one pinned host area; rank 0's finalize kernel writes the other
      void *ds = nullptr, *dout = nullptr;
      MHIP(rk, hipHostGetDevicePointer(&ds, M->h_in, 0));
      d_src = static_cast<const float*>(ds); d_tgt = d_src + (size_t)n * 3;
      if (r == 0) {
        MHIP(rk, hipHostGetDevicePointer(&dout, M->h_out, 0));
        d_Rt = static_cast<float*>(dout); d_mask = static_cast<uint8_t*>(dout) + 64;
      }
    } else {
      if ((e = grow(rk, &rk.src, &rk.src_cap, (size_t)n * 12))) return e;
      if ((e = grow(rk, &rk.tgt, &rk.tgt_cap, (size_t)n * 12))) return e;
      MHIP(rk, hipMemcpyAsync(rk.src, M->src, (size_t)n * 12, hipMemcpyHostToDevice, rk.stream));
      MHIP(rk, hipMemcpyAsync(rk.tgt, M->tgt, (size_t)n * 12, hipMemcpyHostToDevice, rk.stream));
      d_src = static_cast<const float*>(rk.src); d_tgt = static_cast<const float*>(rk.tgt);
    }
    if (!d_mask) {  // the other ranks (and big inputs) finalize into device memory of their own
      if ((e = grow(rk, &rk.mask, &rk.mask_cap, (size_t)n))) return e;
      d_Rt = rk.Rt; d_mask = static_cast<uint8_t*>(rk.mask);
    }
    return SC_OK;
  };
  if (replicated) {
    compute([&]() -> int {
      int e = setup_io(false);
      if (e) return e;
      sc_stats st; memset(&st, 0, sizeof st); st.size = sizeof st;
      e = sc_hypothesize_device(rk.ctx, d_src, d_tgt, n, &p, rk.keys + 2 * r, &st);  // A, B for the whole job; C1 + C2 on this rank's blocks
      return e ? fail_from_ctx(e) : SC_OK;
    });
  } else {
  compute([&]() -> int {
    int e = setup_io(true);
    if (e) return e;
    e = sc_shard_compat_device(rk.ctx, d_src, d_tgt, n, &p, rk.bits);
    return e ? fail_from_ctx(e) : SC_OK;
  });
  collective([&] { return rccl_allgather(M, r, rk.bits, plan.bits_bytes_per_rank); },
             [&](bool ok) { return loop_allgather(M, r, ok, [](Rank& q) { return q.bits; }, plan.bits_bytes_per_rank); });
  compute([&]() -> int { const int e = sc_shard_edges_device(rk.ctx, rk.hist); return e ? fail_from_ctx(e) : SC_OK; });
  if (!est)  // (SC_FLAG_EST_BOUND: every rank took the whole sample — one collective fewer; `est` is the same on every rank)
    collective([&] { return rccl_allreduce_hist(M, r); }, [&](bool ok) { return loop_allreduce_hist(M, r, ok); });
  compute([&]() -> int {
    const int e = sc_shard_select_device(rk.ctx, rk.hist, static_cast<char*>(rk.cand) + (size_t)r * plan.cand_bytes_per_rank);
    return e ? fail_from_ctx(e) : SC_OK;
  });
  collective([&] { return rccl_allgather(M, r, rk.cand, plan.cand_bytes_per_rank); },
             [&](bool ok) { return loop_allgather(M, r, ok, [](Rank& q) { return q.cand; }, plan.cand_bytes_per_rank); });
  compute([&]() -> int {
    sc_stats st; memset(&st, 0, sizeof st); st.size = sizeof st;
    const int e = sc_shard_score_device(rk.ctx, rk.cand, rk.keys + 2 * r, &st);
    return e ? fail_from_ctx(e) : SC_OK;
  });
  }
  collective([&] { return rccl_allgather(M, r, rk.keys, 16); },   // the key pairs: 16 bytes per rank
             [&](bool ok) { return loop_allgather(M, r, ok, [](Rank& q) { return static_cast<void*>(q.keys); }, 16); });
  int fin = SC_OK;
  compute([&]() -> int {
    memset(&rk.stats, 0, sizeof rk.stats); rk.stats.size = sizeof rk.stats;
    fin = sc_finalize_gathered_device(rk.ctx, rk.keys, G, d_Rt, d_mask, &rk.stats);
    if (fin == SC_ERETRY || fin == SC_EBOUND) { MHIP(rk, hipStreamSynchronize(rk.stream)); return SC_OK; }  // every rank alike: the caller re-runs
    if (fin != SC_OK && fin != SC_ENOHYP) return fail_from_ctx(fin);
    if (r == 0 && !M->pinned_io) {  // rank 0 returns the outputs (every rank holds the same ones)
      float Rt[12];
      MHIP(rk, hipMemcpyAsync(Rt, rk.Rt, 48, hipMemcpyDeviceToHost, rk.stream));
      MHIP(rk, hipMemcpyAsync(M->mask, rk.mask, (size_t)n, hipMemcpyDeviceToHost, rk.stream));
      MHIP(rk, hipStreamSynchronize(rk.stream));
      memcpy(M->R, Rt, 36); memcpy(M->t, Rt + 9, 12);
    } else {
      MHIP(rk, hipStreamSynchronize(rk.stream));  // (pinned outputs: the caller's thread copies them out of h_out)
    }
    return SC_OK;
  });
  return rc != SC_OK ? rc : fin;
}

void worker_main(sc_multi* M, int r) {
  uint64_t seen = 0;
  for (;;) {
    // wait for a job: spin for ~200 us (calls in a row: a registration loop), then sleep
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    bool have = false;
    for (;;) {
      if (M->quit.load(std::memory_order_acquire)) return;
      if (M->job.load(std::memory_order_acquire) != seen) { have = true; break; }
      spin_pause();
      if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
    if (!have) {
      std::unique_lock<std::mutex> lk(M->m);
      M->sleepers.fetch_add(1);
      M->cv_go.wait(lk, [&] { return M->quit.load() || M->job.load() != seen; });
      M->sleepers.fetch_sub(1);
      if (M->quit.load()) return;
    }
    seen = M->job.load(std::memory_order_acquire);
    M->ranks[r].error.clear();
    M->ranks[r].status = run_rank(M, r);
    M->running.fetch_sub(1, std::memory_order_acq_rel);
  }
}

// wake the workers for job number job + 1 and wait until all of them are done with it
void run_job(sc_multi* M) {
  M->bar.worst.store(SC_OK, std::memory_order_relaxed);
  M->running.store(M->n, std::memory_order_relaxed);
  M->job.fetch_add(1);  // seq_cst: ordered against the sleepers count below (no lost wake-up)
  if (M->sleepers.load() > 0) {
    { std::lock_guard<std::mutex> lk(M->m); }  // a worker between its check and its wait holds the mutex
    M->cv_go.notify_all();
  }
  spin_until([&] { return M->running.load(std::memory_order_acquire) == 0; });
}

int grow_pinned(sc_multi* M, void** p, size_t* cap, size_t want) {
  if (*cap >= want) return SC_OK;
  if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }  // (no call in flight: the workers are idle)
  const size_t sz = want + want / 4 + 4096;
  if (hipHostMalloc(p, sz, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError(); *p = nullptr; M->last_error = "hipHostMalloc failed (staging area)"; return SC_ENOMEM;
  }
  *cap = sz;
  return SC_OK;
}

int create_common(const int* device_ids, int n_dev, bool loopback, sc_multi** out) {
  if (!out) return SC_EINVAL;
  *out = nullptr;
  if (!device_ids || n_dev < 1 || n_dev > 64) return SC_EINVAL;
  if (!loopback)
    for (int a = 0; a < n_dev; a++)
      for (int b = a + 1; b < n_dev; b++)
        if (device_ids[a] == device_ids[b]) return SC_EINVAL;  // one rank per device
  sc_multi* M = new (std::nothrow) sc_multi();
  if (!M) return SC_ENOMEM;
  // loopback with ONE rank: the rank machinery over a real single-rank RCCL communicator — the one way to execute the
  // RCCL calls themselves (dlopen, ncclCommInitAll, ncclAllGather, ncclAllReduce) on a one-GPU box
  const bool rccl_single = loopback && n_dev == 1;
  if (rccl_single) loopback = false;
  M->n = n_dev; M->loopback = loopback; M->workers = n_dev > 1 || rccl_single;
  M->ranks.resize(n_dev);
  M->bar.n = n_dev;
  for (int r = 0; r < n_dev; r++) {
    Rank& rk = M->ranks[r];
    rk.device = device_ids[r];
    const int rc = sc_create(rk.device, &rk.ctx);
    if (rc) { sc_destroy_multi(M); return rc; }
    if (M->workers) {
      if (hipSetDevice(rk.device) != hipSuccess || hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&rk.hist), SC_HIST_WORDS * 4) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&rk.keys), 16 * (size_t)n_dev) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&rk.Rt), 64) != hipSuccess) { sc_destroy_multi(M); return SC_EHIP; }
      (void)sc_set_stream(rk.ctx, rk.stream);  // the collectives are ordered with the kernels on this stream
    }
  }
  if (M->workers && !loopback) {  // sc_create_multi with n_dev == 1 never touches RCCL
    if (!M->rccl.open(M->last_error)) { sc_destroy_multi(M); return SC_ERCCL; }
    std::vector<ncclComm_t> comms(n_dev);
    const ncclResult_t e = M->rccl.CommInitAll(comms.data(), n_dev, device_ids);
    if (e != ncclSuccess) { sc_destroy_multi(M); return SC_ERCCL; }
    for (int r = 0; r < n_dev; r++) M->ranks[r].comm = comms[r];
  }
  if (M->workers)
    for (int r = 0; r < n_dev; r++) M->ranks[r].worker = std::thread(worker_main, M, r);
  *out = M;
  return SC_OK;
}

}  // namespace

extern "C" {

int sc_create_multi(const int* device_ids, int n_dev, sc_multi** out) { return create_common(device_ids, n_dev, false, out); }

int sc_create_multi_loopback(int device, int n_ranks, sc_multi** out) {
  if (n_ranks < 1 || n_ranks > 64) return SC_EINVAL;
  std::vector<int> ids((size_t)n_ranks, device);
  return create_common(ids.data(), n_ranks, true, out);
}

void sc_destroy_multi(sc_multi* M) {
  if (!M) return;
  {
    std::lock_guard<std::mutex> lk(M->m);
    M->quit.store(true);
  }
  M->cv_go.notify_all();
  for (Rank& rk : M->ranks) if (rk.worker.joinable()) rk.worker.join();
  for (Rank& rk : M->ranks) {
    (void)hipSetDevice(rk.device);
    if (rk.stream && !M->broken.load()) (void)hipStreamSynchronize(rk.stream);
    if (rk.comm && M->rccl.CommDestroy) (void)M->rccl.CommDestroy(rk.comm);
    if (rk.ctx) { (void)sc_set_stream(rk.ctx, nullptr); sc_destroy(rk.ctx); }
    for (void* p : {rk.bits, rk.cand, rk.src, rk.tgt, rk.mask, (void*)rk.hist, (void*)rk.keys, (void*)rk.Rt})
      if (p) (void)hipFree(p);
    if (rk.stream) (void)hipStreamDestroy(rk.stream);
  }
  if (M->h_in) (void)hipHostFree(M->h_in);
  if (M->h_out) (void)hipHostFree(M->h_out);
  if (M->rccl.lib) dlclose(M->rccl.lib);
  delete M;
}

const char* sc_multi_last_error(const sc_multi* M) { return M ? M->last_error.c_str() : "null handle"; }

int sc_register_multi(sc_multi* M, const float* src, const float* tgt, int64_t n, const sc_params* p, float R[9],
                      float t[3], uint8_t* mask, sc_stats* stats) {
  if (!M || !src || !tgt || !p || !R || !t || !mask || n < 3 || n > (1 << 24)) return SC_EINVAL;
  if (p->size != sizeof(sc_params) || p->shard_world != 1) return SC_EINVAL;  // the sharding is this call's business
  if (!M->workers) return sc_register(M->ranks[0].ctx, src, tgt, n, p, R, t, mask, stats);  // no RCCL call at all
  if (M->broken.load()) { M->last_error = "an earlier collective failed and the communicators were aborted: create a new handle"; return SC_ERCCL; }
  M->src = src; M->tgt = tgt; M->npts = n; M->params = *p; M->R = R; M->t = t; M->mask = mask;
  M->pinned_io = n <= PINNED_MAX_N;
  if (M->pinned_io) {  // two memcpys on the caller's thread instead of 2 x n_dev pageable-memory copies
    (void)hipSetDevice(M->ranks[0].device);
    int e = grow_pinned(M, &M->h_in, &M->h_in_cap, (size_t)n * 24);
    if (e == SC_OK) e = grow_pinned(M, &M->h_out, &M->h_out_cap, 64 + (size_t)n);
    if (e != SC_OK) return e;
    memcpy(M->h_in, src, (size_t)n * 12);
    memcpy(static_cast<char*>(M->h_in) + (size_t)n * 12, tgt, (size_t)n * 12);
  }
  for (int attempt = 0;; attempt++) {
    run_job(M);
    // SC_ERETRY: a candidate blob was too small for this input.  Every rank sees the same blobs and reports it together;
    // bigger blobs from now on (sticky), and the call runs again.
    bool retry = true, bound = false;
    for (int r = 0; r < M->n; r++) { retry = retry && M->ranks[r].status == SC_ERETRY; bound = bound || M->ranks[r].status == SC_EBOUND; }
    // SC_EBOUND from ANY rank (a rank whose context has another history — recreated, warmed up differently — can fail
    // the validation of a host-free enqueue alone): every rank runs the call again, without the flag this once.  Two reasons hide
    // behind the status: the ESTIMATE was too high (the same on every rank; sc_debug_last.prune_bound == 2) — after two of those
    // the handle certifies for good — or a host-free call's covers were outgrown, which says nothing about the next frame.
    if (bound && !M->certify_once && attempt < 16) {
      bool est_failed = false;
      for (int r = 0; r < M->n; r++) {
        sc_debug_info di; memset(&di, 0, sizeof di); di.size = sizeof di;
        if (M->ranks[r].status == SC_EBOUND && sc_debug_last(M->ranks[r].ctx, &di) == SC_OK && di.prune_bound == 2) est_failed = true;
      }
      if (est_failed && ++M->est_fails >= 2) M->estimate = false;
      M->certify_once = true;
      continue;
    }
    M->certify_once = false;
    if (!retry || attempt >= 16) break;
    M->cand_level++;
  }
  int rc = SC_OK;
  for (int r = 0; r < M->n; r++) {
    const int s = M->ranks[r].status;
    if (s != SC_OK && s != SC_ENOHYP && rc == SC_OK) {
      rc = s;
      M->last_error = "rank " + std::to_string(r) + " (device " + std::to_string(M->ranks[r].device) + "): " + M->ranks[r].error;
    }
  }
  if (rc != SC_OK) return rc;
  if (M->pinned_io) {
    const float* Rt = static_cast<const float*>(M->h_out);
    memcpy(R, Rt, 36); memcpy(t, Rt + 9, 12);
    memcpy(mask, static_cast<const uint8_t*>(M->h_out) + 64, (size_t)n);
  }
  if (stats && stats->size == sizeof(sc_stats)) {
    *stats = M->ranks[0].stats;
    uint64_t ws = 0; uint32_t scored = 0;
    for (const Rank& rk : M->ranks) { ws += rk.stats.workspace_bytes; scored += rk.stats.tri_scored; }
    stats->workspace_bytes = ws;   // over all devices
    stats->tri_scored = scored;    // hypotheses scored by all ranks (= tri_kept)
  }
  return M->ranks[0].status;  // SC_OK or SC_ENOHYP: identical on every rank
}

}  // extern "C"
