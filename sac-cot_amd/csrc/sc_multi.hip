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
//  * a LOOPBACK transport (sc_create_multi_loopback: n ranks on ONE device, device copies between the ranks' buffers
//    behind host barriers) exists so that the whole orchestration — buffers, phase order, error agreement, outputs —
//    is exercised bit for bit on a one-GPU box.  Only the RCCL calls themselves are then untested.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: every function is resolved with dlsym

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/saccot.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
    return true;
  }
};

// sense-reversing barrier of the worker threads; also agrees on the worst status so far
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int n = 0, waiting = 0;
  uint64_t gen = 0;
  int worst = SC_OK;
  int arrive(int status) {  // returns the worst status any worker has reported this call
    std::unique_lock<std::mutex> lk(m);
    if (status != SC_OK && worst == SC_OK) worst = status;
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
    return worst;
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
  std::vector<Rank> ranks;
  Rccl rccl;
  Barrier bar;
  std::string last_error;
  // job hand-off to the workers
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  uint64_t job = 0;
  int running = 0;
  bool quit = false;
  // the call in flight
  const float *src = nullptr, *tgt = nullptr;
  int64_t npts = 0;
  sc_params params{};
  int cand_level = 0;  // sticky: raised whenever a call came back with SC_ERETRY (candidate blobs too small)
  float* R = nullptr; float* t = nullptr; uint8_t* mask = nullptr;
  std::vector<uint32_t> hist_host;  // loopback all-reduce
};

namespace {

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

// ---- transports: in-place all-gather of `per` bytes per rank, and SUM all-reduce of n u32 ----------------------
int xfer_allgather(sc_multi* M, int r, void* Rank::*buf, size_t per) {
  Rank& rk = M->ranks[r];
  char* mine = static_cast<char*>(rk.*buf);
  if (!M->loopback) {
    const ncclResult_t e = M->rccl.AllGather(mine + (size_t)r * per, mine, per, ncclUint8, rk.comm, rk.stream);
    if (e != ncclSuccess) { rk.error = std::string("ncclAllGather: ") + M->rccl.GetErrorString(e); return SC_ERCCL; }
    return SC_OK;
  }
  // loopback: every rank's slice must exist before anyone copies it, and nobody may move on (and overwrite) earlier
  MHIP(rk, hipStreamSynchronize(rk.stream));
  M->bar.arrive(SC_OK);
  for (int q = 0; q < M->n; q++)
    if (q != r)
      MHIP(rk, hipMemcpyAsync(mine + (size_t)q * per, static_cast<char*>(M->ranks[q].*buf) + (size_t)q * per, per,
                              hipMemcpyDeviceToDevice, rk.stream));
  MHIP(rk, hipStreamSynchronize(rk.stream));
  M->bar.arrive(SC_OK);
  return SC_OK;
}

int xfer_allreduce_hist(sc_multi* M, int r) {
  Rank& rk = M->ranks[r];
  if (!M->loopback) {
    const ncclResult_t e = M->rccl.AllReduce(rk.hist, rk.hist, SC_HIST_WORDS, ncclUint32, ncclSum, rk.comm, rk.stream);
    if (e != ncclSuccess) { rk.error = std::string("ncclAllReduce: ") + M->rccl.GetErrorString(e); return SC_ERCCL; }
    return SC_OK;
  }
  MHIP(rk, hipStreamSynchronize(rk.stream));
  M->bar.arrive(SC_OK);
  std::vector<uint32_t> sum(SC_HIST_WORDS, 0u), one(SC_HIST_WORDS);
  for (int q = 0; q < M->n; q++) {
    MHIP(rk, hipMemcpy(one.data(), M->ranks[q].hist, SC_HIST_WORDS * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < SC_HIST_WORDS; b++) sum[b] += one[b];
  }
  M->bar.arrive(SC_OK);  // everyone has read every histogram before anyone overwrites its own
  MHIP(rk, hipMemcpy(rk.hist, sum.data(), SC_HIST_WORDS * 4, hipMemcpyHostToDevice));
  return SC_OK;
}

// One rank's part of a call.  Every rank reaches the same barriers in the same order whatever happens: a rank that
// has failed keeps arriving (with its status) and skips the work, and nobody enters a collective unless every rank
// got there without an error — a collective one rank never joins would hang the others.
int run_rank(sc_multi* M, int r) {
  Rank& rk = M->ranks[r];
  const int G = M->n;
  const int64_t n = M->npts;
  sc_params p = M->params;
  p.shard_rank = r; p.shard_world = G;
  p.shard_cand_level = M->cand_level;
  if (p.shard_block == 0) p.shard_block = 1024;
  int rc = SC_OK;
  sc_shard_plan plan; plan.size = sizeof plan;
  auto fail_from_ctx = [&](int code) { rk.error = sc_last_error(rk.ctx); return code; };
  auto phase = [&](auto&& body) {  // run `body` unless somebody failed; then agree
    if (rc == SC_OK) rc = body();
    const int worst = M->bar.arrive(rc);
    if (rc == SC_OK && worst != SC_OK) rc = worst;  // another rank failed: stop here too (status of the first failure)
  };
  phase([&]() -> int {
    MHIP(rk, hipSetDevice(rk.device));
    int e = sc_shard_plan_query(&p, n, &plan);
    if (e) { rk.error = "bad parameters"; return e; }
    if ((e = grow(rk, &rk.bits, &rk.bits_cap, plan.bits_bytes_total))) return e;
    if ((e = grow(rk, &rk.cand, &rk.cand_cap, (size_t)G * plan.cand_bytes_per_rank))) return e;
    if ((e = grow(rk, &rk.src, &rk.src_cap, (size_t)n * 12))) return e;
    if ((e = grow(rk, &rk.tgt, &rk.tgt_cap, (size_t)n * 12))) return e;
    if ((e = grow(rk, &rk.mask, &rk.mask_cap, (size_t)n))) return e;
    MHIP(rk, hipMemcpyAsync(rk.src, M->src, (size_t)n * 12, hipMemcpyHostToDevice, rk.stream));
    MHIP(rk, hipMemcpyAsync(rk.tgt, M->tgt, (size_t)n * 12, hipMemcpyHostToDevice, rk.stream));
    e = sc_shard_compat_device(rk.ctx, static_cast<const float*>(rk.src), static_cast<const float*>(rk.tgt), n, &p, rk.bits);
    return e ? fail_from_ctx(e) : SC_OK;
  });
  phase([&]() -> int { return xfer_allgather(M, r, &Rank::bits, plan.bits_bytes_per_rank); });
  phase([&]() -> int { const int e = sc_shard_edges_device(rk.ctx, rk.hist); return e ? fail_from_ctx(e) : SC_OK; });
  phase([&]() -> int { return xfer_allreduce_hist(M, r); });
  phase([&]() -> int {
    const int e = sc_shard_select_device(rk.ctx, rk.hist, static_cast<char*>(rk.cand) + (size_t)r * plan.cand_bytes_per_rank);
    return e ? fail_from_ctx(e) : SC_OK;
  });
  phase([&]() -> int { return xfer_allgather(M, r, &Rank::cand, plan.cand_bytes_per_rank); });
  phase([&]() -> int {
    sc_stats st; memset(&st, 0, sizeof st); st.size = sizeof st;
    const int e = sc_shard_score_device(rk.ctx, rk.cand, rk.keys + 2 * r, &st);
    return e ? fail_from_ctx(e) : SC_OK;
  });
  phase([&]() -> int {
    // the key pairs: same in-place all-gather, 16 bytes per rank
    Rank& me = rk;
    char* mine = reinterpret_cast<char*>(me.keys);
    if (!M->loopback) {
      const ncclResult_t e = M->rccl.AllGather(mine + 16 * (size_t)r, mine, 16, ncclUint8, me.comm, me.stream);
      if (e != ncclSuccess) { me.error = std::string("ncclAllGather: ") + M->rccl.GetErrorString(e); return SC_ERCCL; }
      return SC_OK;
    }
    MHIP(me, hipStreamSynchronize(me.stream));
    M->bar.arrive(SC_OK);
    for (int q = 0; q < G; q++)
      if (q != r)
        MHIP(me, hipMemcpyAsync(mine + 16 * (size_t)q, reinterpret_cast<char*>(M->ranks[q].keys) + 16 * (size_t)q, 16,
                                hipMemcpyDeviceToDevice, me.stream));
    MHIP(me, hipStreamSynchronize(me.stream));
    M->bar.arrive(SC_OK);
    return SC_OK;
  });
  int fin = SC_OK;
  phase([&]() -> int {
    memset(&rk.stats, 0, sizeof rk.stats); rk.stats.size = sizeof rk.stats;
    fin = sc_finalize_gathered_device(rk.ctx, rk.keys, G, rk.Rt, static_cast<uint8_t*>(rk.mask), &rk.stats);
    if (fin == SC_ERETRY) { MHIP(rk, hipStreamSynchronize(rk.stream)); return SC_OK; }  // every rank alike: the caller re-runs
    if (fin != SC_OK && fin != SC_ENOHYP) return fail_from_ctx(fin);
    if (r == 0) {  // rank 0 returns the outputs (every rank holds the same ones)
      float Rt[12];
      MHIP(rk, hipMemcpyAsync(Rt, rk.Rt, 48, hipMemcpyDeviceToHost, rk.stream));
      MHIP(rk, hipMemcpyAsync(M->mask, rk.mask, (size_t)n, hipMemcpyDeviceToHost, rk.stream));
      MHIP(rk, hipStreamSynchronize(rk.stream));
      memcpy(M->R, Rt, 36); memcpy(M->t, Rt + 9, 12);
    } else {
      MHIP(rk, hipStreamSynchronize(rk.stream));
    }
    return SC_OK;
  });
  return rc != SC_OK ? rc : fin;
}

void worker_main(sc_multi* M, int r) {
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(M->m);
      M->cv_go.wait(lk, [&] { return M->quit || M->job != seen; });
      if (M->quit) return;
      seen = M->job;
    }
    M->ranks[r].error.clear();
    M->ranks[r].status = run_rank(M, r);
    {
      std::lock_guard<std::mutex> lk(M->m);
      if (--M->running == 0) M->cv_done.notify_all();
    }
  }
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
    M->quit = true;
  }
  M->cv_go.notify_all();
  for (Rank& rk : M->ranks) if (rk.worker.joinable()) rk.worker.join();
  for (Rank& rk : M->ranks) {
    (void)hipSetDevice(rk.device);
    if (rk.stream) (void)hipStreamSynchronize(rk.stream);
    if (rk.comm && M->rccl.CommDestroy) (void)M->rccl.CommDestroy(rk.comm);
    if (rk.ctx) { (void)sc_set_stream(rk.ctx, nullptr); sc_destroy(rk.ctx); }
    for (void* p : {rk.bits, rk.cand, rk.src, rk.tgt, rk.mask, (void*)rk.hist, (void*)rk.keys, (void*)rk.Rt})
      if (p) (void)hipFree(p);
    if (rk.stream) (void)hipStreamDestroy(rk.stream);
  }
  if (M->rccl.lib) dlclose(M->rccl.lib);
  delete M;
}

const char* sc_multi_last_error(const sc_multi* M) { return M ? M->last_error.c_str() : "null handle"; }

int sc_register_multi(sc_multi* M, const float* src, const float* tgt, int64_t n, const sc_params* p, float R[9],
                      float t[3], uint8_t* mask, sc_stats* stats) {
  if (!M || !src || !tgt || !p || !R || !t || !mask || n < 3 || n > (1 << 24)) return SC_EINVAL;
  if (p->size != sizeof(sc_params) || p->shard_world != 1) return SC_EINVAL;  // the sharding is this call's business
  if (!M->workers) return sc_register(M->ranks[0].ctx, src, tgt, n, p, R, t, mask, stats);  // no RCCL call at all
  M->src = src; M->tgt = tgt; M->npts = n; M->params = *p; M->R = R; M->t = t; M->mask = mask;
  for (int attempt = 0;; attempt++) {
    M->bar.worst = SC_OK;
    {
      std::lock_guard<std::mutex> lk(M->m);
      M->running = M->n;
      M->job++;
    }
    M->cv_go.notify_all();
    {
      std::unique_lock<std::mutex> lk(M->m);
      M->cv_done.wait(lk, [&] { return M->running == 0; });
    }
    // SC_ERETRY: a candidate blob was too small for this input.  Every rank sees the same blobs and reports it together;
    // bigger blobs from now on (sticky), and the call runs again.
    bool retry = true;
    for (int r = 0; r < M->n; r++) retry = retry && M->ranks[r].status == SC_ERETRY;
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
