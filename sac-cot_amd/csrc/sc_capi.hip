// sc_capi.hip — the C ABI of include/saccot.h: context, grow-only device workspace, stage sequencing.
//
// There is no reference interface to mirror (the reference tree is /root/reference/README.md:1-2); the
// boundary is SURVEY.md §8(b).  Everything that computes runs on the GPU: this file only validates,
// allocates, enqueues kernels (sc_kernels.hpp) and copies.  Two small read-backs (edge count, triangle
// count) size the data-dependent buffers; there is no CPU fallback of any stage.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/saccot.h"
#include "../../include/saccot_debug.h"
#include "sc_kernels.hpp"
#include "sc_gramref.hpp"

using namespace sc;

namespace {

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  template <class T> T* as() const { return static_cast<T*>(p); }
};

constexpr int N_EVENTS = 12;  // 0..8 stage brackets, 9..10 the key kernel (all reused by calibrate_events), 11 the stop of stage C2's filter kernel (SC_FLAG_TIMING_HOT)
constexpr int N_PINNED = 32;  // [16..21]: the bounding boxes of the two clouds (stage_points_kernel)

}  // namespace

struct sc_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string last_error;
  size_t held = 0;
  uint64_t cap_bytes = 64ull << 30;
  hipEvent_t ev[N_EVENTS] = {};
  uint64_t* pinned = nullptr;  // N_PINNED x u64 host-pinned area the kernels write results into

  // workspace
  Buf in_src, in_tgt, planes, S, bits, deg, degp, wpre, ebase, edge_off, scan_tmp, ei, ej, es, ebi, ebj, tcnt, toff, wkey, kcol, ctl, events, blk_gt,
      blk_eq, blk_minmax, bits2, off_gt, off_eq, sel_ord, sel_key, sortkey, sorted, sort_tmp, tri, tri_rk, key_rk, rt, rt_aos, partial, cnt, key, rt12,
      mask, refine_tmp, amx_pairs, strong, rowcost, cost_pre, lb_state, lb_ticket, fx_tile, fx_state, fx_mx, fx_part, fx_coef, guard_tmp, fx_frame, ref_cand;
  // the XCD-aware block orders of stage A (compat_wg_map), one per row width met so far: a context that alternates between a few
  // sizes must not rebuild and upload the map on every call (that cost 2 ms per call in bench.py's varying-n leg)
  static constexpr int N_WG_MAPS = 8;
  struct WgMap { int W = 0; uint32_t len = 0; Buf buf; uint64_t used = 0; } wg_maps[N_WG_MAPS];
  uint64_t wg_map_clock = 0;
  bool filter_on = false;   // C2 of the running / last call goes through a matrix-pipe filter (decided ONCE per call)
  int filter_mode = 0;      // ... which: 1 linear, 2 Gram (0: the plain fp32 kernel)
  FilterPlan fx_plan{};     // ... with this plan (sc_debug_last reads the filter's counters through it)
  // sc_register (host arrays in, host arrays out): pinned, device-mapped staging areas — the staging kernel reads the
  // correspondences straight from host memory and the finalize kernel writes (R, t, mask) straight into it: no copies
  void* h_in = nullptr; size_t h_in_cap = 0;
  void* h_out = nullptr; size_t h_out_cap = 0;

  // state of the last hypothesize call (consumed by finalize)
  int n = 0, ld = 0;
  uint64_t E = 0, M = 0, M_total = 0;
  uint64_t ev_capacity = 1ull << 21;  // event records (32 B each); doubled after an overflow
  bool pruned = false, use_events = false, have_total = false;
  uint32_t T_eff = 0;
  Derived dv{};
  Shard sh{};
  bool have_hyp = false, begun = false;
  sc_params params{};  // the parameters of the running call (begin -> end)
  bool timing = false, timing_hot = false;
  uint32_t amx_blocks = 0;  // workgroups of the last arg-max launch whose pairs this context's finalize step reduces itself (0: one reduced pair in `key`)
  bool hot_ext = false;  // SC_FLAG_TIMING_HOT took its timestamps from the kernels' dispatch packets (run_stage_c)
  int timing_one = -1;  // SC_FLAG_TIMING_ONE: the one stage bracket recorded this call (0 .. 6), -1: none
  float ev_overhead_us = -1.f;  // cost of one event record inside a bracket (calibrate_events); < 0: not measured yet
  bool timed_trikeys = false;
  bool refine = false;
  const uint64_t* mbits = nullptr;
  const float* smin_ptr = nullptr;
  Tuning tn;  // defaults unless sc_set_debug() changed them; the library reads no environment variable
  // the adjacency bit matrix of the running call: the context's own buffer, or — sharded stage A — the caller's
  // all-gathered one
  uint64_t* bits_cur = nullptr;
  // sharded A + B (sc_shard_*_device): phase reached (0: none), the gathered candidate blobs
  // decoupled look-back launches (single-pass scan, fused compaction): their state area and its epoch
  uint32_t lb_epoch = 0;
  void* lb_zeroed = nullptr;
  size_t lb_zeroed_cap = 0;
  bool rows_fused = false;  // run_row_stats already produced edge_off / ebase / cost_pre and armed the edge count
  // ---- host-free enqueue (sc_register_device_async; see include/saccot.h).  A call of the same shape as the last one does
  // not wait for stage B's two counts: its launches cover E_cov edges / M_cov keys and read the real counts from device
  // memory; sc_wait validates (spec_validate) and repeats the call the waiting way when a count outgrew the cover.
  bool spec_on = false;      // the running call is enqueued host-free
  bool fast_ok = false;      // the last completed call was regular (events, a-priori window, T triangles found): the next may try
  int fast_state = 0;        // last call: 0 waited, 1 host-free and valid, 2 host-free, failed validation, repeated
  uint64_t E_cov = 0, M_cov = 0, E_last = 0, M_last = 0;
  int last_n = 0;
  sc_params last_p{};
  // r05: what the covers are sized by.  A stream of DIFFERENT frames of one shape (bench.py's: 32 scenes whose inlier ratio moves the
  // edge count by 1.6 x and the triangle count by 4 x) outgrew "the last call's counts plus half" in a frame out of six; the covers now
  // follow the largest counts of the last HI_WINDOW .. 2 HI_WINDOW completed calls of the shape (two buckets: the current one and the one
  // before it), so a stream pays for the spread of its frames once.  A change of shape empties the window (note_completed).
  static constexpr uint32_t HI_WINDOW = 64, HI_YOUNG = 8;
  uint64_t E_hi[2] = {0, 0}, M_hi[2] = {0, 0};
  uint32_t hi_n = 0, hi_seen = 0;  // calls in the current bucket; regular calls of this shape seen so far (saturating)
  // cumulative over the context's life (sc_debug_last): how its sc_register_device(_async) / host-free sc_hypothesize_device calls
  // were enqueued and how stage B's pruning bound fared — what a stream of frames reports (a per-call field would only say the last)
  uint64_t n_frames = 0, n_fast_ok = 0, n_fast_repeat = 0, n_est_ok = 0, n_est_fail = 0;
  uint64_t n_spec_grow = 0;  // buffers re-allocated (a stream synchronisation each) inside host-free enqueues
  // the coordinate maxima and boxes the staging kernel of the last completed call published (use_filter): what a host-free call
  // picks stage C2's kernel by when it is enqueued before its own staging kernel has run — a second frame in flight on the stream
  uint64_t mx_last = ~0ull, box_last[6] = {0, 0, 0, 0, 0, 0};
  // the outstanding call of sc_register_device_async (at most one per context)
  bool regular = false;      // the running / last call met every assumption of the host-free form (run_select, finalize_wait)
  bool pending = false;
  bool pend_done = false;    // ... and it was a waited call: complete, status in pending_rc
  int pending_rc = 0;        // status already known when the async half returned (a waited call is complete by then)
  bool pend_finalize = false;  // the outstanding call is sc_finalize_gathered_device_async's (sc_wait: no repeat inside the library)
  const float* pend_src = nullptr; const float* pend_tgt = nullptr; float* pend_Rt = nullptr; uint8_t* pend_mask = nullptr;
  int64_t pend_n = 0; sc_params pend_p{}; sc_stats pend_stats{};
  // pruning by an ESTIMATED bound (sc_tri.hip 3c): only where this file can repeat the call itself (sc_register_device /
  // _async / sc_register), never through the phase API; the select verifies the bound, finalize_wait reads the verdict
  bool est_allowed = false;  // the running call came in through an entry point that can repeat it
  bool est_active = false;   // ... and prunes by an estimate
  bool est_void = false;     // ... which could not be verified on the path taken (event overflow): repeat
  bool est_failed = false;   // an estimate failed on this context: it certifies for the next est_holdoff completed calls (sc_set_debug resets)
  // r04c: not for ever.  One frame whose estimate fails — a change of scene — used to cost the context its estimating sample (25 us per
  // C2 call) for the rest of its life; now the k-th failure costs 64 << min(k - 1, 6) certifying calls, then the context estimates
  // again: a stream whose estimates always fail wastes one repeated call in 4096.
  uint32_t est_holdoff = 0, est_failures = 0;
  int est_state = 0;         // last pass: 0 certified bound (or no pruning), 1 estimated and verified
  bool est_failed_call = false;  // the running / last call saw its estimate fail and was repeated (sc_debug_last: prune_bound 2)
  SamplePlan plan{false, 1u, 0};
  // stage C2's reference frame (sc_gramref.hpp): on the hot path the estimating sample leaves candidate triangles behind
  // (ref_cand_n of them) and the counting pass carries the vote as an extra workgroup (ref_done); everywhere else stage C
  // votes in a launch of its own
  uint32_t ref_cand_n = 0;
  bool ref_done = false;
  bool build = false;        // the running call takes launch_edge_build (row statistics + edge list + estimating sample in one launch)
  // run-time probe of the matrix pipe's accumulation model (sc_score.hip gram_guard): 0 not run, 1 holds, 2 violated
  int gram_guard = 0;
  float gram_guard_worst = 0.f;
  bool sharded_ab = false;
  int shard_phase = 0;
  const void* cand_all = nullptr;
  size_t cand_bytes = 0;
};

namespace {

int fail_hip(sc_ctx* c, hipError_t e, const char* what) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  if (c) c->last_error = buf;
  return SC_EHIP;
}

#define HIPCHK(c, expr)                                        \
  do {                                                         \
    hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) return fail_hip((c), _e, #expr);     \
  } while (0)

int ensure(sc_ctx* c, Buf& b, size_t bytes) {
  if (bytes <= b.cap) return SC_OK;
  size_t want = bytes + bytes / 8 + 256;
  // a host-free call that has to grow a buffer stalls (the re-allocation below synchronises the stream): grow in big steps there,
  // so that a stream whose covers are still rising pays a few of these, not one per frame; and never allocate crumbs
  if (c->spec_on) want = 2 * bytes + 256;
  if (want < 65536) want = 65536;
  if (c->held - b.cap + want > c->cap_bytes) {
    want = bytes;
    if (c->held - b.cap + want > c->cap_bytes) { c->last_error = "workspace cap exceeded"; return SC_ENOMEM; }
  }
  if (b.p) {
    if (c->spec_on) c->n_spec_grow++;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(b.p));
    c->held -= b.cap;
    b.p = nullptr; b.cap = 0;
  }
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { b.p = nullptr; (void)hipGetLastError(); c->last_error = "hipMalloc failed"; return SC_ENOMEM; }
  b.cap = want;
  c->held += want;
  return SC_OK;
}

#define ENSURE(c, buf, bytes)                      \
  do {                                             \
    int _rc = ensure((c), (buf), (bytes));         \
    if (_rc != SC_OK) return _rc;                  \
  } while (0)

// `need` bytes now, `room` bytes wanted (room for the counts of the NEXT calls of this shape: a host-free call's launches are
// sized by a cover of the recent counts and must fit the arrays it finds — fast_plan); under a tight workspace cap: just `need`
int ensure_room(sc_ctx* c, Buf& b, size_t need, size_t room) {
  if (room > need && room > b.cap) {
    const size_t want = room + room / 8 + 256;
    if (c->held - b.cap + want <= c->cap_bytes && ensure(c, b, room) == SC_OK) return SC_OK;
  }
  return ensure(c, b, need);
}
#define ENSURE_ROOM(c, buf, need, room)                        \
  do {                                                         \
    int _rc = ensure_room((c), (buf), (need), (room));         \
    if (_rc != SC_OK) return _rc;                              \
  } while (0)

// what a host-free repetition of a call's shape is sized to cover (fast_plan): the last count plus half, and the largest count of
// the shape's recent calls plus a quarter (sc_ctx::E_hi / M_hi)
// While the shape is young on this context (fewer than HI_YOUNG of its calls seen) the last count is doubled instead: the first
// frames of a stream say little about the spread of the frames to come, and a miss costs a whole repeated call (bench.py's
// stream, frames of 386 k .. 697 k edges: with "plus half" from the first frame on, one of the first twenty was repeated).
uint64_t cover_of(uint64_t last, const uint64_t hi[2], bool young) {
  const uint64_t h = hi[0] > hi[1] ? hi[0] : hi[1];
  const uint64_t b = h + h / 4;
  // young: twice the last count (or the window's maximum plus a quarter, if larger); later the window alone decides — the last
  // count is in it — so that a stream's covers stay put from frame to frame instead of following every large frame by half
  // (a cover beyond 2^20 edges takes the scan's two-launch form, and every launch sized by it grows with it)
  const uint64_t a = young ? 2 * last : (h ? 0 : last + last / 2);
  return (a > b ? a : b) + 4096;
}

// entries the edge arrays hold (what fast_plan caps a host-free call's edge cover by)
uint64_t edge_capacity(const sc_ctx* c, bool build) {
  uint64_t cap = c->es.cap / 4 >= 2 ? c->es.cap / 4 - 2 : 0;
  for (const Buf* b : {&c->ei, &c->ej}) cap = b->cap / 4 < cap ? b->cap / 4 : cap;
  if (!build) for (const Buf* b : {&c->ebi, &c->ebj}) cap = b->cap / 4 < cap ? b->cap / 4 : cap;
  return cap;
}
bool same_shape(const sc_params* p, const sc_params* q);
// entries a waited call leaves in an array indexed by one of stage B's two counts: what the host-free repetitions of its shape
// will want to cover (the window's maxima only while the shape continues: note_completed empties it after a change)
uint64_t room_of(const sc_ctx* c, uint64_t count, const uint64_t hi[2]) {
  static const uint64_t none[2] = {0, 0};
  const bool continues = c->n == c->last_n && same_shape(&c->params, &c->last_p);
  return cover_of(count, continues ? hi : none, !continues || c->hi_seen < sc_ctx::HI_YOUNG);
}

int check_params(const sc_params* p) {
  if (!p || p->size != sizeof(sc_params)) return SC_EINVAL;
  if (!(p->sigma > 0.f) || !(p->t_cmp > 0.f) || !(p->t_cmp < 1.f) || !(p->tau > 0.f) || !(p->min_len >= 0.f))
    return SC_EINVAL;
  if (!std::isfinite(p->sigma) || !std::isfinite(p->tau) || !std::isfinite(p->min_len)) return SC_EINVAL;
  if (p->max_triangles == 0 || p->max_triangles > 0xFFFFFF00u) return SC_EINVAL;  // padded to 256 in u32
  if (p->rank_mode != SC_RANK_WEIGHT && p->rank_mode != SC_RANK_DEGREE) return SC_EINVAL;
  if (p->layout != SC_AOS && p->layout != SC_SOA) return SC_EINVAL;
  if (p->shard_world < 1 || p->shard_rank < 0 || p->shard_rank >= p->shard_world) return SC_EINVAL;
  if (p->score_mode < SC_SCORE_COUNT || p->score_mode > SC_SCORE_MAE) return SC_EINVAL;
  if (p->shard_cand_level < -16 || p->shard_cand_level > 30) return SC_EINVAL;
  return SC_OK;
}

// thresholds precomputed on the host in fp64, rounded once to fp32
Derived derive(const sc_params* p) {
  Derived d;
  d.d_thr = (float)((double)p->sigma * std::sqrt(-2.0 * std::log((double)p->t_cmp)));
  d.neg_inv2sig2 = (float)(-1.0 / (2.0 * (double)p->sigma * (double)p->sigma));
  d.tau2 = (float)((double)p->tau * (double)p->tau);
  d.min_len = p->min_len;
  d.inv_tau2 = (float)(1.0 / ((double)p->tau * (double)p->tau));
  d.inv_tau = (float)(1.0 / (double)p->tau);
  return d;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

Points points_of(const sc_ctx* c) { return Points{c->planes.as<float>(), c->n, c->ld}; }
Graph graph_of(const sc_ctx* c) {
  return Graph{c->bits_cur, c->S.as<float>(), c->deg.as<uint32_t>(), c->degp.as<uint32_t>(),
               c->wpre.as<uint32_t>(), c->n, c->ld, c->ld >> 6};
}
// Look-back launches (single-pass scans, fused row kernel, fused compaction).  lb_state holds the tile descriptors and
// is written by look-back kernels only: zeroed when new, when it grew, and when the 12-bit epoch wraps.  The TICKET and
// error words live in a small buffer of their own (lb_ticket, zeroed once): a ticket inside the descriptor area would,
// after a launch with another layout, sit on an old descriptor and hand out garbage tile indices (that was a GPU
// memory fault at N = 20 000 sharded in development).  slot: which ticket (launches that may be in flight together on
// the stream's order use different ones); desc_off: byte offset of this launch's descriptors inside the area.
int lb_next(sc_ctx* c, size_t bytes, int slot, size_t desc_off, LbArgs* out) {
  ENSURE(c, c->lb_state, bytes);
  if (!c->lb_ticket.p) {
    ENSURE(c, c->lb_ticket, 256);
    HIPCHK(c, hipMemsetAsync(c->lb_ticket.p, 0, c->lb_ticket.cap, c->stream));
  }
  if (c->lb_state.p != c->lb_zeroed || c->lb_state.cap != c->lb_zeroed_cap || c->lb_epoch >= 4095u) {
    HIPCHK(c, hipMemsetAsync(c->lb_state.p, 0, c->lb_state.cap, c->stream));
    c->lb_zeroed = c->lb_state.p; c->lb_zeroed_cap = c->lb_state.cap; c->lb_epoch = 0;
  }
  out->epoch = ++c->lb_epoch;
  out->ticket = c->lb_ticket.as<uint32_t>() + 2 * slot;  // 8 bytes apart
  out->err = c->lb_ticket.as<uint32_t>() + 32 + slot;
  out->desc = reinterpret_cast<uint64_t*>(static_cast<char*>(c->lb_state.p) + desc_off);
  return SC_OK;
}

// how many triangles stage B selects: T — or, when B is sharded, what one candidate blob holds (this rank's own best)
uint32_t select_want(const sc_ctx* c, const sc_params* p) {
  if (!c->sharded_ab) return p->max_triangles;
  return (uint32_t)cand_cap(p->max_triangles, (uint32_t)p->shard_world, p->shard_cand_level) < p->max_triangles
             ? (uint32_t)cand_cap(p->max_triangles, (uint32_t)p->shard_world, p->shard_cand_level)
             : p->max_triangles;
}

// sharded stage B: the device-side edge range [lo, hi) this rank enumerates (nullptr: every edge)
const uint64_t* own_range_of(const sc_ctx* c) {
  return c->sharded_ab ? c->ctl.as<ControlBlock>()->own_edge : nullptr;
}

// event i of the per-stage timing; SC_FLAG_TIMING_HOT keeps only the bracket of the dominant (score) kernel
TriSource tri_source_of(const sc_ctx* c) {
  TriSource ts{c->sel_ord.as<uint64_t>(), c->kcol.as<uint2>(), c->ei.as<uint32_t>(), c->ej.as<uint32_t>(), nullptr, 0, 0, 0u, 0u, 0ull};
  if (c->spec_on) { ts.lim_vertex = (uint32_t)c->n; ts.lim_edge = (uint32_t)c->E_cov; ts.lim_ord = c->M_cov; }
  // an ESTIMATED pruning bound that turns out too high leaves a selection shorter than the T_eff the launches were sized for
  // (the call is then repeated): what lies beyond it in sel_ord must not be followed
  else if (c->est_active) { ts.lim_vertex = (uint32_t)c->n; ts.lim_edge = (uint32_t)c->E; ts.lim_ord = c->M; }
  if (c->sharded_ab && c->cand_all) {  // the selection indexes the gathered candidate blobs
    const size_t cap = cand_cap(c->params.max_triangles, (uint32_t)c->params.shard_world, c->params.shard_cand_level);
    ts.cand_recs = cand_blob(const_cast<void*>(c->cand_all), cap).recs;
    ts.cand_seg = cap;
    ts.cand_stride = c->cand_bytes / 16;
  }
  return ts;
}

// stage brackets as (first event, second event): stage, compat, triangles, kabsch, score, argmax, mask
constexpr int STAGE_EV[7][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {5, 6}, {7, 8}};

int rec(sc_ctx* c, int i) {
  const bool hot = i == 4 || i == 5;
  const bool one = c->timing_one >= 0 && (i == STAGE_EV[c->timing_one][0] || i == STAGE_EV[c->timing_one][1]);
  if (c->timing || (c->timing_hot && hot) || one) HIPCHK(c, hipEventRecord(c->ev[i], c->stream));
  return SC_OK;
}

int calibrate_events(sc_ctx* c);
bool fast_plan(sc_ctx* c, int64_t n, const sc_params* p);

// SC_FLAG_TIMING / _HOT / _ONE -> the context's timing state for this call
int set_timing(sc_ctx* c, const sc_params* p) {
  c->timing = (p->flags & SC_FLAG_TIMING) != 0;
  c->timing_hot = !c->timing && (p->flags & SC_FLAG_TIMING_HOT) != 0;
  c->timing_one = -1;
  if (!c->timing && !c->timing_hot && (p->flags & SC_FLAG_TIMING_ONE)) {
    const int stage = (int)((p->flags >> 8) & 15u);
    if (stage > 6) return SC_EINVAL;
    c->timing_one = stage;
  }
  if (c->timing || c->timing_hot || c->timing_one >= 0) return calibrate_events(c);
  return SC_OK;
}

// user points (device) -> padded planes; zero the "non-finite" flag first
int stage_inputs(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p) {
  if (n < 3 || n > (1 << 24)) return SC_EINVAL;
  if (p->score_mode != SC_SCORE_COUNT && n > (1 << 21)) {  // 1024 n must stay below 2^31 (the score rides a u32)
    c->last_error = "score_mode MSE / MAE needs n <= 2^21";
    return SC_EINVAL;
  }
  c->n = (int)n;
  c->ld = round_up((int)n, 64);
  ENSURE(c, c->planes, (size_t)(6 + 8) * c->ld * sizeof(float));  // 6 SoA planes + the AoS copy (8 floats each)
  // per-call control block (flags, histograms, counters, select state): cleared by the staging kernel itself
  ENSURE(c, c->ctl, sizeof(ControlBlock));
  static_assert(sizeof(ControlBlock) % 4 == 0, "cleared word-wise");
  c->pinned[1] = 0;  // "non-finite input" flag lives in host-pinned memory: the kernel only touches it on bad data
  if (!c->fx_mx.p) {  // coordinate statistics for C2's filters (written whole by every call's staging kernel) + its ticket
    ENSURE(c, c->fx_mx, (FX_MX_WORDS + 1) * 4);
    HIPCHK(c, hipMemsetAsync(c->fx_mx.p, 0, (FX_MX_WORDS + 1) * 4, c->stream));
  }
  ENSURE(c, c->fx_part, stage_part_words(c->ld) * 4);
  c->pinned[13] = ~0ull;  // "maxima not known yet"
  if (c->build) ENSURE(c, c->degp, (size_t)c->ld * sizeof(uint32_t));  // stage A accumulates deg+ there: cleared on the way
  launch_stage_points(d_src, d_tgt, c->n, c->ld, p->layout, c->planes.as<float>(),
                      reinterpret_cast<uint32_t*>(&c->pinned[1]), c->ctl.as<uint32_t>(),
                      (uint32_t)(sizeof(ControlBlock) / 4), c->fx_mx.as<uint32_t>(), c->fx_part.as<uint32_t>(),
                      c->spec_on ? nullptr : c->fx_mx.as<uint32_t>() + FX_MX_WORDS, c->spec_on ? nullptr : &c->pinned[13], c->spec_on ? nullptr : &c->pinned[16], c->stream,  // (host-free: no ticket — stage A's launch reduces the rows, run_compat; finalize hands the words over, DeferredPub)
                      c->build ? c->degp.as<uint32_t>() : nullptr, c->build ? (uint32_t)c->ld : 0u);
  return SC_OK;
}

// dense: also write the n x n weight matrix S (SC_FLAG_NO_DENSE_S clears it: nothing after stage A reads S)
// (r02 / r03 experiment, removed in r04: S from a second, low-priority stream while stage B runs — slower on every config,
// DESIGN.md §5 "measured and dropped")
int run_compat(sc_ctx* c, bool dense) {
  const size_t n = c->n, ld = c->ld, W = ld >> 6;
  if (dense) ENSURE(c, c->S, n * ld * sizeof(float));
  ENSURE(c, c->bits, n * W * sizeof(uint64_t));
  ENSURE(c, c->deg, n * sizeof(uint32_t));
  ENSURE(c, c->degp, ld * sizeof(uint32_t));
  ENSURE(c, c->wpre, n * W * sizeof(uint32_t));
  c->bits_cur = c->bits.as<uint64_t>();
  c->sharded_ab = false; c->shard_phase = 0; c->cand_all = nullptr;
  const uint32_t* map = nullptr;
  uint32_t map_len = 0;
  // The XCD-aware block order (sc_compat.hip compat_wg_map; made once per row width).  Measured r04 (profiles/r04_pmc_compat_xcd_order.txt):
  // HBM write bytes C2 115.8 -> 106.9 MB (algorithmic 103.2: 1.035 x), bits only 13.0 -> 5.1 MB; C3 1820 -> 1683 MB (1.02 x).  Time:
  // C2 25.0 -> 23.7 us, but at C3 the index order is FASTER (352 vs 369 - 389 us, alternating three times in one process) although it
  // writes more: each XCD then streams its S rows into one 512 x 512 corner of the matrix at a time.  So: by size, like the tile height.
  if (!c->tn.compat_linear_order && c->tn.compat_rows != 64 && (c->n < 10000 || c->tn.compat_rows == 16)) {
    sc_ctx::WgMap* slot = nullptr;
    for (sc_ctx::WgMap& m : c->wg_maps) if (m.W == (int)W) slot = &m;
    if (!slot) {  // a width not met before (or evicted): the least recently used slot
      slot = &c->wg_maps[0];
      for (sc_ctx::WgMap& m : c->wg_maps) if (m.used < slot->used) slot = &m;
      const std::vector<uint32_t> m = compat_wg_map((int)W);
      ENSURE(c, slot->buf, m.size() * 4);
      HIPCHK(c, hipMemcpyAsync(slot->buf.p, m.data(), m.size() * 4, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));  // (m is a local: the copy must have left it; once per size)
      slot->W = (int)W; slot->len = (uint32_t)m.size();
    }
    slot->used = ++c->wg_map_clock;
    map = slot->buf.as<uint32_t>(); map_len = slot->len;
  }
  launch_compat(points_of(c), c->dv, dense ? c->S.as<float>() : nullptr, c->bits_cur, 0, c->n, c->tn, c->stream,
                c->build ? c->degp.as<uint32_t>() : nullptr, map, map_len, c->spec_on ? c->fx_part.as<uint32_t>() : nullptr,
                c->fx_mx.as<uint32_t>());
  return SC_OK;
}

// deg / deg+ / word-prefix popcounts from the bit rows (first kernel of stage B's timing bracket)
void arm_word(sc_ctx* c, int idx);

// hot: the fused form (row statistics + CSR offsets + edge count in one launch; run_edges then launches no scan)
int run_row_stats(sc_ctx* c, bool will_prune, bool hot = false) {
  c->rows_fused = false;
  if (hot && c->build) {  // launch_edge_build (run_edges) does all of it
    ENSURE(c, c->bits2, (size_t)c->n * (c->ld >> 6) * sizeof(uint64_t));
    return SC_OK;
  }
  uint64_t* zero_rows = nullptr;
  if (will_prune) {  // the pruned bit matrix is cleared on the way (no separate memset)
    ENSURE(c, c->bits2, (size_t)c->n * (c->ld >> 6) * sizeof(uint64_t));
    zero_rows = c->bits2.as<uint64_t>();
  }
  uint32_t* rowcost = nullptr;
  if (c->sharded_ab) {  // per-row work estimate: its prefix splits the rows between the ranks
    ENSURE(c, c->rowcost, ((size_t)c->n + 4 + 1024) * 4);  // (cost_split_kernel reads whole 16-byte pieces)
    rowcost = c->rowcost.as<uint32_t>();
  }
  // The fused form pays while the look-back stays shallow: 157 tiles at N = 5000 (12.7 us against 5.5 + 7.5 and a launch
  // gap, and edge_fill gets the precomputed bases: 15.8 -> 14.6); at N = 20 000 its 625 tiles cost 61 us against ~40.
  if (hot && !c->tn.rows_unfused && c->n <= 8192) {
    const size_t n = c->n;
    ENSURE(c, c->edge_off, (n + 1) * sizeof(uint64_t));
    ENSURE(c, c->ebase, n * 4);
    uint64_t* cost_pre = nullptr;
    if (c->sharded_ab) { ENSURE(c, c->cost_pre, (n + 1) * sizeof(uint64_t)); cost_pre = c->cost_pre.as<uint64_t>(); }
    LbArgs lb;
    { const int lrc = lb_next(c, row_stats_scan_state_bytes(c->n), 0, 0, &lb); if (lrc) return lrc; }
    arm_word(c, 0);  // read-back #1 (the edge count) is published by this kernel
    launch_row_stats_scan(points_of(c), c->bits_cur, c->deg.as<uint32_t>(), c->degp.as<uint32_t>(), c->wpre.as<uint32_t>(),
                          zero_rows, c->edge_off.as<uint64_t>(), c->ebase.as<uint32_t>(), cost_pre, lb, &c->pinned[0],
                          c->stream);
    c->rows_fused = true;
    return SC_OK;
  }
  launch_row_stats(points_of(c), c->bits_cur, c->deg.as<uint32_t>(), c->degp.as<uint32_t>(),
                   c->wpre.as<uint32_t>(), zero_rows, rowcost, c->stream);
  return SC_OK;
}

// Read-backs: the producing kernel stores its 8-byte result into host-pinned memory (publish_host) and the host
// polls that word instead of blocking in hipStreamSynchronize (whose wake-up alone costs ~10 us, twice per call).
// After ~1 ms of polling it falls back to the blocking wait (very long stages, or a failed launch).
constexpr uint64_t PIN_PENDING = ~0ull;
inline void cpu_relax() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  __builtin_ia32_pause();
#endif
}
void arm_word(sc_ctx* c, int idx) { c->pinned[idx] = PIN_PENDING; }
int wait_word(sc_ctx* c, int idx) {
  volatile uint64_t* w = &c->pinned[idx];
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (*w == PIN_PENDING) {
    cpu_relax();
    if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1000)) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (*w == PIN_PENDING) { c->last_error = "a kernel did not deliver its result"; return SC_EHIP; }
  return SC_OK;
}

bool may_prune(const sc_params* p) { return p->rank_mode == SC_RANK_WEIGHT && !(p->flags & SC_FLAG_NO_PRUNE); }

// The hot path's fused form of row_stats_scan + edge_fill + the estimating sample (sc_tri.hip 1'): where the pruning bound
// will be an estimate (the same conditions as sample_plan's in run_edges, minus what only the edge count decides) and the
// bit rows are short enough for the kernel
bool edge_build_ok(const sc_ctx* c, const sc_params* p, int64_t n) {
  const bool window_known = p->rank_mode == SC_RANK_WEIGHT && 3.0f * p->t_cmp * 0.999f >= 2.0f;
  return c->est_allowed && !c->est_failed && window_known && may_prune(p) && !c->tn.no_events && !c->tn.no_estimate &&
         c->tn.sample_mode == 0 && !c->tn.no_edge_build && !c->tn.rows_unfused && edge_build_fits((int)n) &&  // (stage C may be dealt over ranks: shard_world)
         p->max_triangles != 0;
}

// stage B; on return c->E, c->M, c->T_eff are set and tri/trikey hold the ranked list
// Stage B, first half: CSR edge list and this process's share (`part` of `parts`) of the pruning sample, accumulated
// into `hist` (256 u32 on the device, already zero; nullptr = the control block's own histogram).  Sets c->E and the
// decisions the second half needs.
// the Gram filter's frame (sc_gramref.hpp): 512 bytes per context
int ensure_frame(sc_ctx* c) {
  if (!c->fx_frame.p) {
    ENSURE(c, c->fx_frame, gram_frame_bytes());
    HIPCHK(c, hipMemsetAsync(c->fx_frame.p, 0, gram_frame_bytes(), c->stream));
  }
  return SC_OK;
}

int run_edges(sc_ctx* c, const sc_params* p, uint32_t* hist, uint32_t part, uint32_t parts) {
  const size_t n = c->n;
  hipStream_t st = c->stream;
  c->ref_cand_n = 0; c->ref_done = false;
  ENSURE(c, c->edge_off, (n + 1) * sizeof(uint64_t));
  ENSURE(c, c->scan_tmp, scan_temp_bytes(n));
  // read-back #1: the scan kernel itself writes the edge count to host-pinned memory (no copy kernel)
  const bool build = c->build && hist == nullptr && parts == 1;
  if (c->build && !build) { c->last_error = "internal: the fused edge kernel was chosen for a call that shares its sample"; return SC_EHIP; }
  const bool fused = c->rows_fused || build;
  c->rows_fused = false;
  if (!fused || build) arm_word(c, 0);
  ENSURE(c, c->ebase, n * 4);
  ScanExtra xe;  // the scan of deg+ also writes the per-row CSR bases edge_fill reads
  xe.deg = c->deg.as<uint32_t>(); xe.degp = c->degp.as<uint32_t>(); xe.ebase = c->ebase.as<uint32_t>();
  ScanExtra xc;  // (sharded) the scan of the row costs
  if (!fused && scan_writes_ebase(n)) {  // tiled scans: the single-pass form, each on its own half of the state area
    const size_t half = scan_temp_bytes(n);
    { const int lrc = lb_next(c, 2 * half, 0, 0, &xe.lb); if (lrc) return lrc; }
    if (c->sharded_ab) { const int lrc = lb_next(c, 2 * half, 1, half, &xc.lb); if (lrc) return lrc; }  // (same size: the area stays)
  }
  if (c->sharded_ab) {
    // also the prefix of the per-row work estimate and, from it, this rank's contiguous row / edge range
    ENSURE(c, c->cost_pre, (n + 1) * sizeof(uint64_t));
    if (!fused)
      launch_scan_u32_pair(c->degp.as<uint32_t>(), c->edge_off.as<uint64_t>(), c->rowcost.as<uint32_t>(),
                           c->cost_pre.as<uint64_t>(), n, c->scan_tmp.p, c->tn, st, &c->pinned[0], &xe, &xc);
    ControlBlock* ctl = c->ctl.as<ControlBlock>();
    launch_shard_split(c->cost_pre.as<uint64_t>(), c->edge_off.as<uint64_t>(), c->n, (uint32_t)p->shard_rank,
                       (uint32_t)p->shard_world, ctl->own_row, ctl->own_edge, st);
  } else if (!fused) {
    launch_scan_u32(c->degp.as<uint32_t>(), n, c->edge_off.as<uint64_t>(), c->scan_tmp.p, c->tn, st, &c->pinned[0], &xe);
  }
  // While the host polls for the edge count, edge_fill already runs into the edge arrays this context holds from
  // earlier calls (it needs no host-side count: one wave per row, offsets from the scan).  Writes beyond their
  // capacity are dropped by the kernel; in that case, or on a first call, it runs (again) after the read-back.
  const Graph g = graph_of(c);
  uint64_t spec_cap = c->es.cap / 4 >= 2 ? c->es.cap / 4 - 2 : 0;  // es carries two pad entries
  for (const Buf* b : {&c->ei, &c->ej}) spec_cap = b->cap / 4 < spec_cap ? b->cap / 4 : spec_cap;
  if (!build) for (const Buf* b : {&c->ebi, &c->ebj}) spec_cap = b->cap / 4 < spec_cap ? b->cap / 4 : spec_cap;
  // the weight histogram of the heaviest-edge pruning sample is collected by the same kernel (sc_debug.sample_mode = ...
  // any: it costs edge_fill ~1 us and saves a launch whenever that sample is chosen)
  uint32_t* es_hist = c->ctl.as<ControlBlock>()->es_hist;
  auto fill_edges = [&](uint64_t cap) {
    if (build) {
      launch_edge_build(g, points_of(c), c->dv, c->bits2.as<uint64_t>(), c->edge_off.as<uint64_t>(), c->ebase.as<uint32_t>(),
                        c->ei.as<uint32_t>(), c->ej.as<uint32_t>(), c->es.as<float>(), cap, c->spec_on ? nullptr : &c->pinned[0], st,
                        c->spec_on ? c->ctl.as<ControlBlock>()->live_edges : nullptr);
      return;
    }
    launch_edge_fill(g, points_of(c), c->dv, c->edge_off.as<uint64_t>(), c->ei.as<uint32_t>(), c->ej.as<uint32_t>(),
                     c->es.as<float>(), c->ebase.as<uint32_t>(), fused || scan_writes_ebase(n), c->ebi.as<uint32_t>(),
                     c->ebj.as<uint32_t>(), cap, es_hist, st);
  };
  if (spec_cap || build) fill_edges(spec_cap);  // (the fused kernel also makes the row statistics and the count itself)
  // Host-free call (c->spec_on; fast_plan() made sure E_cov <= spec_cap): no wait.  E is then what the launches and arrays
  // COVER; the kernels below take the real count from edge_off[n] (E_dev) and the end of the call validates it.
  const bool spec = c->spec_on;
  const uint64_t* E_dev = spec ? c->edge_off.as<uint64_t>() + n : nullptr;
  if (!spec) {
    { const int wrc = wait_word(c, 0); if (wrc) return wrc; }
    if ((uint32_t)c->pinned[1] != 0) { c->last_error = "non-finite input coordinate"; return SC_EINVAL; }
  }
  const uint64_t E = c->E = spec ? c->E_cov : c->pinned[0];
  c->M = 0; c->M_total = 0; c->T_eff = 0; c->pruned = false; c->use_events = false; c->have_total = false;
  c->est_active = false; c->est_void = false; c->plan = SamplePlan{false, 1u, p->max_triangles};
  c->pinned[14] = 0;  // "the select found fewer keys above the pruning bound than it promised" (select_round_kernel)
  // an exchanged histogram is written on every path (zeros where no sample runs: the control block's copies are zero)
  if (hist && (E == 0 || !(may_prune(p) && E >= 4096))) launch_hist_reduce(c->ctl.as<ControlBlock>()->prune_hist, hist, st);
  if (E == 0) return SC_OK;
  // edge ids, CSR bases and the strong list are u32 (include/saccot.h, limits): a graph beyond that is refused, not wrapped
  if (E >= (1ull << 32)) { c->last_error = "the compatibility graph has 2^32 or more edges"; return SC_ETOOMANY; }
  // (a waited call leaves room for what a host-free repetition of its shape will want to cover: fast_plan caps by these arrays)
  const uint64_t E_room = spec ? E : room_of(c, E, c->E_hi);
  const size_t edge_arrays_before[5] = {c->ei.cap, c->ej.cap, c->es.cap, c->ebi.cap, c->ebj.cap};
  ENSURE_ROOM(c, c->ei, E * 4, E_room * 4);
  ENSURE_ROOM(c, c->ej, E * 4, E_room * 4);
  ENSURE_ROOM(c, c->es, (E + 2) * 4, (E_room + 2) * 4);  // +1: the 4-way unrolled gathers of idle slots may touch index E
  if (!build) { ENSURE_ROOM(c, c->ebi, E * 4, E_room * 4); ENSURE_ROOM(c, c->ebj, E * 4, E_room * 4); }
  // every other array indexed by the edge count gets room for whatever the edge arrays can hold — that is what fast_plan caps a
  // host-free call's cover by, so no such call ever has to grow one of them (a re-allocation synchronises the stream: 0.35 ms in
  // the middle of bench.py's stream, once per array and rise of the cover)
  const uint64_t E_hold = spec ? E : edge_capacity(c, build);
  ENSURE_ROOM(c, c->tcnt, E * 4, E_hold * 4);
  ENSURE_ROOM(c, c->toff, (E + 1) * 8, (E_hold + 1) * 8);
  ENSURE_ROOM(c, c->scan_tmp, scan_temp_bytes(E), scan_temp_bytes(E_hold));
  if (!spec) {
    ENSURE_ROOM(c, c->strong, strong_list_bytes(E), strong_list_bytes(E_hold));
    ENSURE_ROOM(c, c->lb_state, scan_temp_bytes(E), 2 * scan_temp_bytes(E_hold));
  }
  // (an array that was re-allocated above — for this call's count, or only for the ROOM the next calls want — has lost what the
  // speculative launch wrote into it.  By CAPACITY, not by address: hipFree + hipMalloc hand the same address back often enough)
  const size_t edge_arrays_after[5] = {c->ei.cap, c->ej.cap, c->es.cap, c->ebi.cap, c->ebj.cap};
  bool edge_arrays_moved = false;
  for (int k = 0; k < (build ? 3 : 5); k++) edge_arrays_moved = edge_arrays_moved || edge_arrays_before[k] != edge_arrays_after[k];
  if (E > spec_cap || edge_arrays_moved) {  // first call, or the graph outgrew the arrays (just re-allocated above)
    if (spec_cap && !build) HIPCHK(c, hipMemsetAsync(es_hist, 0, sizeof(uint32_t) * PR_HCOPIES * 256, st));  // the partial run's counts
    fill_edges(E);
  }
  // certified pruning (sc_tri.hip 3b): weight ranking only; pointless on tiny graphs
  c->pruned = may_prune(p) && E >= 4096;
  // counting pass + event list (sc_tri.hip 2b) on the pruned graph; Tuning::no_events keeps the row-walking pair
  c->use_events = c->pruned && !c->tn.no_events;
  if (c->pruned) {
    ControlBlock* ctl = c->ctl.as<ControlBlock>();
    if (p->flags & SC_FLAG_EXACT_TOTAL) {  // statistics only: 3-cliques of the whole graph
      launch_tri_count(g, g.bits, c->es.as<float>(), nullptr, c->ei.as<uint32_t>(), c->ej.as<uint32_t>(), E,
                       c->tcnt.as<uint32_t>(), nullptr, c->tn, st);
      launch_scan_u32(c->tcnt.as<uint32_t>(), E, c->toff.as<uint64_t>(), c->scan_tmp.p, c->tn, st, &c->pinned[4]);
      c->have_total = true;
    }
    // the smallest possible weight is ~3 t_cmp (every edge has s >= t_cmp up to rounding); 0.1 % slack
    // An ESTIMATED bound where the call can be repeated should the select find it too high (sc_tri.hip 3c): the common
    // form only — event list, a-priori select window (the check rides the window's first round), the whole sample here
    const bool window_known = p->rank_mode == SC_RANK_WEIGHT && 3.0f * p->t_cmp * 0.999f >= 2.0f;
    // (sharded, SC_FLAG_EST_BOUND: every rank takes the WHOLE sample — it is cheaper than the latency of the all-reduce that
    // would sum the ranks' shares — and the merge of the candidates verifies the bound: sc_shard_score_device)
    const bool est_local = c->est_allowed && !c->est_failed && hist == nullptr && parts == 1 && !c->sharded_ab;
    const bool est_shard = c->sharded_ab && hist != nullptr && (p->flags & SC_FLAG_EST_BOUND) != 0;
    c->plan = sample_plan(p->max_triangles, (est_local || est_shard) && c->use_events && window_known, c->tn,
                          spec ? c->E_last : E, g.W);
    c->est_active = c->plan.estimate;
    if (build && !c->plan.estimate) { c->last_error = "internal: the fused edge kernel ran but the bound is not an estimate"; return SC_EHIP; }
    if (c->plan.estimate) {
      // an estimating sample (the single-GPU hot path; sharded, SC_FLAG_EST_BOUND: every rank takes the whole of it), inlier count, a call
      // big enough for stage C2's Gram filter to be in question: the sample also
      // leaves its workgroups' best triangles behind — the voters of that filter's reference frame (run_select passes them on)
      uint4* cand = nullptr;
      if ((est_local || est_shard) && p->score_mode == 0 && c->tn.score_filter != 1 && c->tn.score_filter != 2 && !c->tn.gram_ref_late &&
          (uint64_t)p->max_triangles * (uint64_t)c->n >= (1ull << 27)) {
        if (!c->ref_cand.p) {
          ENSURE(c, c->ref_cand, 4096 * sizeof(uint4));
          HIPCHK(c, hipMemsetAsync(c->ref_cand.p, 0, 4096 * sizeof(uint4), st));
        }
        cand = c->ref_cand.as<uint4>();
        c->ref_cand_n = sample_candidate_blocks(E, c->tn);
      }
      launch_sample_estimate(g, build ? nullptr : c->ebi.as<uint32_t>(), build ? nullptr : c->ebj.as<uint32_t>(), c->ei.as<uint32_t>(),
                             c->ej.as<uint32_t>(), c->es.as<float>(), E, 3.0f * p->t_cmp * 0.999f, c->plan.rate, ctl->prune_hist, c->tn,
                             st, E_dev, c->ebase.as<uint32_t>(), cand, cand ? ctl->ref_slot : nullptr);
    } else {
      // SC_FLAG_EST_BOUND on the phase API where the plan is NOT an estimate (select window unknown: t_cmp below ~0.668; sc_debug's
      // no_events / no_estimate / sample_mode): the caller still skips the histogram all-reduce, so every rank must end up with the
      // SAME histogram — each takes the WHOLE certifying sample, not its share (with shares the ranks derived different
      // bounds, cut the strong list differently, and the merged top-T could miss triangles)
      const bool whole = est_shard;
      launch_sample_hist(g, c->ebi.as<uint32_t>(), c->ebj.as<uint32_t>(), c->ei.as<uint32_t>(), c->ej.as<uint32_t>(),
                         c->es.as<float>(), E, p->max_triangles, 3.0f * p->t_cmp * 0.999f, whole ? 0u : part, whole ? 1u : parts,
                         ctl->prune_hist, ctl->es_hist, c->tn, st, E_dev, spec ? c->E_last : 0);
    }
    if (hist) launch_hist_reduce(ctl->prune_hist, hist, st);  // the exchanged form: one 256-bin histogram
  }
  return SC_OK;
}

// Ordinal-order compaction of the keys at or above the threshold the select found -> sel_ord / sel_key.
// Two launches (count per tile; write, every tile summing the counts before it by itself).  (The ONE-launch form — counts of the
// earlier tiles by decoupled look-back — was built as VERDICT r01 asked, bit-exact and slower: 16.4 us against 4.7 + 4.7 on C2;
// removed in r05.)  Tuning::compact_self_max == 0 (a test) takes the scanned three-launch form.
int run_compaction(sc_ctx* c, const KeyView& view, size_t nb, int rounds, uint64_t* host_short) {
  hipStream_t st = c->stream;
  SelectState* sel = &c->ctl.as<ControlBlock>()->sel;
  launch_compact_count(view, sel, rounds, c->blk_gt.as<uint32_t>(), c->blk_eq.as<uint32_t>(), st, host_short);
  // few tiles (and M < 2^32): compact_write adds up the tile counts itself, no scan launch in between
  const bool self_off = nb <= c->tn.compact_self_max && view.M < (1ull << 32);
  if (!self_off)
    launch_scan_u32_pair(c->blk_gt.as<uint32_t>(), c->off_gt.as<uint64_t>(), c->blk_eq.as<uint32_t>(),
                         c->off_eq.as<uint64_t>(), nb, c->scan_tmp.p, c->tn, st);
  launch_compact_write(view, sel, c->blk_gt.as<uint32_t>(), c->blk_eq.as<uint32_t>(),
                       self_off ? nullptr : c->off_gt.as<uint64_t>(), self_off ? nullptr : c->off_eq.as<uint64_t>(),
                       c->sel_ord.as<uint64_t>(), c->sel_key.as<uint32_t>(),
                       c->sel_ord.cap / 8 < c->sel_key.cap / 4 ? c->sel_ord.cap / 8 : c->sel_key.cap / 4, st);
  return SC_OK;
}

// Stage B, second half: prune with the (summed) sample histogram, enumerate, select.  On return c->M, c->T_eff are set
// and sel_ord / sel_key hold the selection.  want_list: also materialise the T x 3 triangle list (stage hook; the hot
// path reads triangles through TriSource).  hist == nullptr: the control block's own histogram.
int run_select(sc_ctx* c, const sc_params* p, const uint32_t* hist, bool want_list) {
  hipStream_t st = c->stream;
  const uint64_t E = c->E;
  if (E == 0) return SC_OK;
  const bool spec = c->spec_on;  // host-free call: E is the cover, the real count sits in edge_off[n] (see run_edges)
  const uint64_t* E_dev = spec ? c->edge_off.as<uint64_t>() + c->n : nullptr;
  const Graph g = graph_of(c);
  const bool use_events = c->use_events, have_total = c->have_total;
  StrongList sl{nullptr, nullptr, 0};
  const uint64_t* mbits = g.bits;
  const float* smin = nullptr;
  if (c->pruned) {
    ControlBlock* ctl = c->ctl.as<ControlBlock>();
    if (use_events) {  // the pruning kernel also compacts the strong edges for the counting pass
      ENSURE(c, c->strong, strong_list_bytes(E));  // (a waited call: run_edges left room)
      sl = StrongList{c->strong.as<uint32_t>(), ctl->st_fill, strong_list_cap(E), 0u};
    }
    const bool recut = c->sharded_ab && p->shard_world > 1 && use_events;
    // contiguous regions: a rank walks only the regions its own edge range touches (unsharded the modulo form stays: the
    // counting pass is no faster with contiguous regions at C2 / C4 and 20 us slower at C3)
    if (recut) sl.region_blocks = sl.cap / 256;
    // Sharded (event path): the ranks' row ranges are cut AFTER the certificate, by the work of the pruned graph (VERDICT r02
    // #5: a correspondence list in keypoint order puts nearly every strong edge into a few rows).  Every rank holds the
    // whole strong matrix, so every rank computes the same cut: the pruning kernel lists EVERY strong edge, two small
    // launches make the cut, and the counting pass skips the edges of the other ranks.
    launch_prune_bits(g, hist ? hist : ctl->prune_hist, hist == nullptr, c->ei.as<uint32_t>(), c->ej.as<uint32_t>(), c->es.as<float>(), E,
                      c->est_active ? c->plan.hist_want : (uint64_t)p->max_triangles, 3.0f * p->t_cmp * 0.999f, c->bits2.as<uint64_t>(), &ctl->smin, &ctl->klb, sl,
                      c->tcnt.as<uint32_t>(), recut ? nullptr : own_range_of(c), st, E_dev, c->est_active,
                      spec && c->build && !c->sharded_ab);  // (the scan of such a call is trimmed to the real edges: below)
    if (recut) {
      ENSURE(c, c->rowcost, ((size_t)c->n + 4 + 1024) * 4);  // (cost_split_kernel reads whole 16-byte pieces)
      launch_strong_rowcost(g, c->bits2.as<uint64_t>(), c->rowcost.as<uint32_t>(), st);
      launch_cost_split(c->rowcost.as<uint32_t>(), c->edge_off.as<uint64_t>(), c->n, (uint32_t)p->shard_rank,
                        (uint32_t)p->shard_world, ctl->own_row, ctl->own_edge, st);
    }
    mbits = c->bits2.as<uint64_t>();
    smin = &ctl->smin;
  }
  c->mbits = mbits;
  c->smin_ptr = smin;
  EventList ev{};
  if (use_events) {
    // Tuning::event_cap (test hook) forces a (too) small buffer for THIS call without touching the context's own
    // capacity, which only grows (after an overflow)
    const uint64_t ev_cap = c->tn.event_cap >= 256 ? c->tn.event_cap : c->ev_capacity;
    ENSURE(c, c->events, event_bytes(ev_cap));
    c->pinned[5] = 0;
    ev = event_list(c->events.p, ev_cap, g.W, c->ctl.as<ControlBlock>()->ev_fill,
                    reinterpret_cast<uint32_t*>(&c->pinned[5]));
    GramRefJob ref{};
    if (c->ref_cand_n) {  // (run_edges) one extra workgroup votes for stage C2's reference frame under this launch
      int frc = ensure_frame(c);
      if (frc) return frc;
      ref = gram_ref_job(points_of(c), c->fx_mx.as<uint32_t>(), c->dv.tau2, c->tn, c->fx_frame.p);
      ref.src.cand = c->ref_cand.as<uint4>(); ref.src.n_cand = c->ref_cand_n; ref.src.cand_slot = c->ctl.as<ControlBlock>()->ref_slot;
      c->ref_done = true;
    }
    launch_tri_count_events(g, mbits, sl, c->build ? nullptr : c->ebi.as<uint32_t>(), c->build ? nullptr : c->ebj.as<uint32_t>(), c->ei.as<uint32_t>(),
                            c->ej.as<uint32_t>(), spec ? c->E_last : E, p->rank_mode, c->tcnt.as<uint32_t>(), ev, c->tn, st, own_range_of(c),
                            c->ebase.as<uint32_t>(), ref.out ? &ref : nullptr);
  } else {
    launch_tri_count(g, mbits, c->es.as<float>(), smin, c->ei.as<uint32_t>(), c->ej.as<uint32_t>(), E,
                     c->tcnt.as<uint32_t>(), own_range_of(c), c->tn, st);
  }
  // read-back #2: triangle count (of the pruned graph when pruning), written to pinned memory by the scan
  arm_word(c, 2);
  ScanExtra xr;  // sharded: the counts outside this rank's edge range are zero — their tiles are skipped
  xr.range = own_range_of(c);
  // a host-free call on the fused edge kernel: the launches cover E edges, the graph has fewer — the scan skips the tiles beyond
  // them (the pruning kernel's workgroups there have left without writing their counts)
  if (spec && c->build && !c->sharded_ab) xr.range = c->ctl.as<ControlBlock>()->live_edges;
  { const int lrc = lb_next(c, scan_temp_bytes(E), 0, 0, &xr.lb); if (lrc) return lrc; }
  launch_scan_u32(c->tcnt.as<uint32_t>(), E, c->toff.as<uint64_t>(), c->scan_tmp.p, c->tn, st, spec ? nullptr : &c->pinned[2], &xr);
  // While the host polls for the count, the key kernel already runs into the key arrays this context holds from earlier
  // calls (it takes everything else from device memory).  Only in the common form — events, a-priori select window —
  // and not when every stage is bracketed by events; re-run below if the count outgrew the arrays or a region overflowed.
  SelectState* sel = &c->ctl.as<ControlBlock>()->sel;
  const bool window_known = p->rank_mode == SC_RANK_WEIGHT && 3.0f * p->t_cmp * 0.999f >= 2.0f;
  uint64_t spec_cap = 0;
  if (use_events && window_known && !c->timing) {
    spec_cap = c->wkey.cap / 4 < c->kcol.cap / 8 ? c->wkey.cap / 4 : c->kcol.cap / 8;
    if (spec) spec_cap = c->M_cov;  // (<= that capacity: fast_plan)
    if (spec_cap) {
      ENSURE(c, c->blk_minmax, 2 * 8192 * 4);
      launch_tri_keys_events(g, c->es.as<float>(), c->toff.as<uint64_t>(), p->rank_mode, ev, c->wkey.as<uint32_t>(),
                             c->kcol.as<uint2>(), c->blk_minmax.as<uint32_t>(), sel, select_want(c, p),
                             &c->ctl.as<ControlBlock>()->klb, E, spec_cap, c->tn, st, !c->sharded_ab);
    }
  }
  // host-free call: no wait — M is what the key arrays and the launches below cover, T_eff the requested T; the kernels
  // read the real count from toff[E] (= the scan's total: the counts of [real E, E) are zero) and sc_wait validates
  if (!spec) { const int wrc = wait_word(c, 2); if (wrc) return wrc; }
  c->M_total = spec ? c->M_cov : (have_total ? c->pinned[4] : c->pinned[2]);
  const uint64_t M = c->M = spec ? c->M_cov : c->pinned[2];
  if (M == 0) {
    // no triangle in the pruned graph: under a CERTIFIED bound (or none) the graph has none; an ESTIMATED bound has verified nothing
    // (the select that checks it never runs): the call is repeated with a certifying sample (finalize_wait)
    if (c->est_active) c->est_void = true;
    return SC_OK;
  }
  const uint32_t want_sel = select_want(c, p);
  const uint32_t T_eff = c->T_eff = (uint32_t)(M < want_sel ? M : want_sel);
  if (M * 12 > c->cap_bytes) { c->last_error = "triangle keys exceed the workspace cap"; return SC_ETOOMANY; }
  const size_t nb = compact_blocks(M);
  const uint64_t M_room = spec ? M : room_of(c, M, c->M_hi);
  const size_t key_arrays_before[2] = {c->wkey.cap, c->kcol.cap};
  ENSURE_ROOM(c, c->wkey, M * 4, M_room * 4);
  ENSURE_ROOM(c, c->kcol, M * 8, M_room * 8);
  const bool key_arrays_moved = key_arrays_before[0] != c->wkey.cap || key_arrays_before[1] != c->kcol.cap;  // (re-allocated: what the speculative key pass wrote is gone)
  ENSURE(c, c->blk_minmax, 2 * 8192 * 4);
  const size_t nb_room = compact_blocks(spec ? M : (uint64_t)(c->wkey.cap / 4 < c->kcol.cap / 8 ? c->wkey.cap / 4 : c->kcol.cap / 8));  // (whatever the key arrays can hold: fast_plan's cap)
  ENSURE_ROOM(c, c->blk_gt, nb * 4, nb_room * 4);
  ENSURE_ROOM(c, c->blk_eq, nb * 4, nb_room * 4);
  ENSURE_ROOM(c, c->off_gt, (nb + 1) * 8, (nb_room + 1) * 8);
  ENSURE_ROOM(c, c->off_eq, (nb + 1) * 8, (nb_room + 1) * 8);
  ENSURE(c, c->scan_tmp, scan_temp_bytes(nb));
  ENSURE(c, c->sel_ord, (size_t)T_eff * 8);
  ENSURE(c, c->sel_key, (size_t)T_eff * 4);
  if (want_list) ENSURE(c, c->tri, (size_t)T_eff * 12);
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[9], st));
  bool events_ok = use_events;
  if (!spec && use_events && (uint32_t)c->pinned[5] != 0) {  // a region overflowed: this call walks the rows again
    events_ok = false;
    uint64_t want_cap = c->ev_capacity * 2;          // every event holds >= 1 triangle, so M bounds the need
    if (want_cap < M + M / 4) want_cap = M + M / 4;
    if (want_cap > (1ull << 28)) want_cap = 1ull << 28;
    c->ev_capacity = want_cap;
  }
  // weight keys of a graph whose edges all weigh >= 2/3 (0.1 % slack) lie in [2.0, 3.0]: window known a priori
  const bool fast_window = events_ok && window_known;
  const bool keys_done = spec_cap != 0 && events_ok && M <= spec_cap && !key_arrays_moved;  // the speculative pass wrote every key
  if (keys_done) {
    // nothing to do
  } else if (events_ok) {
    launch_tri_keys_events(g, c->es.as<float>(), c->toff.as<uint64_t>(), p->rank_mode, ev, c->wkey.as<uint32_t>(),
                           c->kcol.as<uint2>(), c->blk_minmax.as<uint32_t>(), sel, fast_window ? (uint64_t)select_want(c, p) : (uint64_t)T_eff,
                           fast_window ? &c->ctl.as<ControlBlock>()->klb : nullptr, E, M, c->tn, st, !c->sharded_ab);
  } else {
    launch_tri_keys(g, mbits, smin, c->ebase.as<uint32_t>(), c->ei.as<uint32_t>(), c->ej.as<uint32_t>(),
                    c->es.as<float>(), c->toff.as<uint64_t>(), E, p->rank_mode, c->wkey.as<uint32_t>(),
                    c->kcol.as<uint2>(), c->blk_minmax.as<uint32_t>(), sel, T_eff, own_range_of(c), c->tn, st);
  }
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[10], st));
  c->timed_trikeys = c->timing;
  KeyView view = plain_view(c->wkey.as<uint32_t>(), M);
  if (spec) view.M_dev = c->toff.as<uint64_t>() + E;
  // an estimated bound is verified by the first round over the a-priori window; any other path leaves it unverified
  if (c->est_active && !fast_window) c->est_void = true;
  launch_select_rounds(view, sel, fast_window ? 2 : 3, c->tn, st, c->sharded_ab ? nullptr : &c->pinned[14]);
  { const int crc = run_compaction(c, view, nb, fast_window ? 2 : 3, c->sharded_ab ? nullptr : &c->pinned[14]); if (crc) return crc; }
  // the list stays in ordinal order: no sort on the hot path (the winner is found by (count, key, position))
  if (want_list)
    launch_tri_decode(c->ei.as<uint32_t>(), c->ej.as<uint32_t>(), c->kcol.as<uint2>(), c->sel_ord.as<uint64_t>(), T_eff,
                      c->tri.as<uint32_t>(), st);
  // the shape a host-free repetition of this call assumes (fast_plan / finalize_wait)
  c->regular = !spec && events_ok && fast_window && !have_total && !c->timing && M >= want_sel && hist == nullptr;
  return SC_OK;
}

int run_triangles(sc_ctx* c, const sc_params* p, bool want_list) {
  const int rc = run_edges(c, p, nullptr, 0, 1);
  return rc ? rc : run_select(c, p, nullptr, want_list);
}

void fill_stats(const sc_ctx* c, sc_stats* s) {
  if (!s) return;
  const uint32_t sz = s->size;
  if (sz != sizeof(sc_stats)) return;
  s->n = (uint32_t)c->n;
  s->edges = c->E;
  s->tri_total = c->M_total;
  s->tri_kept = c->T_eff;
  s->tri_scored = c->sh.n_local;
  s->workspace_bytes = c->held;
  // algorithmic bytes of this call (include/saccot.h): SURVEY §8d's per-stage formulas with the call's own counts
  const uint64_t n = (uint64_t)c->n, nn8 = n * n / 8;
  const bool dense = !(c->params.flags & SC_FLAG_NO_DENSE_S);
  s->bytes_moved = (dense ? 4 * n * n : 0) + nn8 + 24 * n          // A: S, bit rows, the correspondences
                   + nn8 + 20 * c->E + 12 * c->M + 16 * (uint64_t)c->T_eff  // B: bit rows once, edge records, keys + {vertex, edge}, selection
                   + 52 * (uint64_t)c->sh.n_local + 24 * n + n;   // C: (R,t) out and in, counts, the correspondences, the mask
}

// A bracket of two event records contains the cost of one record (a barrier packet with a timestamp: ~4-5 us of
// stream time on this part) on top of what lies between them.  That cost is measured once per context — an empty
// bracket on the idle stream — and subtracted, so that a bracket around one kernel reads like the profiler's duration
// of that kernel (plus any genuine launch gap).
int calibrate_events(sc_ctx* c) {
  if (c->ev_overhead_us >= 0.f) return SC_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float best = 1e9f;
  for (int rep = 0; rep < 5; rep++) {
    // a chain of back-to-back records: the first interval includes waking an idle queue, the later ones are the
    // steady cost of a record that follows other work — which is what sits inside a stage bracket
    for (int k = 0; k < 6; k++) HIPCHK(c, hipEventRecord(c->ev[k], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 2; k < 5; k++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]) == hipSuccess && ms * 1000.f < best) best = ms * 1000.f;
    }
  }
  c->ev_overhead_us = best < 1e8f ? best : 0.f;
  return SC_OK;
}

float ev_us_raw(sc_ctx* c, int a, int b) {  // start / stop of kernel dispatches: nothing to subtract
  float ms = 0.f;
  if (hipEventSynchronize(c->ev[b]) != hipSuccess || hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) {
    (void)hipGetLastError();
    return 0.f;
  }
  return ms > 0.f ? ms * 1000.f : 0.f;
}

float ev_us(sc_ctx* c, int a, int b) {
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) { (void)hipGetLastError(); return 0.f; }
  const float us = ms * 1000.f - (c->ev_overhead_us > 0.f ? c->ev_overhead_us : 0.f);
  return us > 0.f ? us : 0.f;
}

int busy(sc_ctx* c) {  // an sc_register_device_async / sc_finalize_gathered_device_async call is outstanding on this context
  if (!c->pending) return SC_OK;
  c->last_error = "a call is outstanding on this context (sc_wait first)";
  return SC_EINVAL;
}

int host_to_planes(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p) {
  { const int brc = busy(c); if (brc) return brc; }
  c->spec_on = false;  // (a host-free call that was never finalized must not leave its covers to a stage hook)
  c->build = false;  // (stage hooks: the separate kernels)
  if (!src || !tgt || n < 3 || n > (1 << 24)) return SC_EINVAL;
  ENSURE(c, c->in_src, (size_t)n * 12);
  ENSURE(c, c->in_tgt, (size_t)n * 12);
  HIPCHK(c, hipMemcpyAsync(c->in_src.p, src, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_tgt.p, tgt, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
  return stage_inputs(c, c->in_src.as<float>(), c->in_tgt.as<float>(), n, p);
}

int check_flag(sc_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if ((uint32_t)c->pinned[1] != 0) { c->last_error = "non-finite input coordinate"; return SC_EINVAL; }
  return SC_OK;
}

}  // namespace

extern "C" {

int sc_version(void) { return (SC_VERSION_MAJOR << 16) | SC_VERSION_MINOR; }

const char* sc_strerror(int status) {
  switch (status) {
    case SC_OK: return "ok";
    case SC_EINVAL: return "invalid argument";
    case SC_ENOMEM: return "out of memory or workspace cap exceeded";
    case SC_EHIP: return "HIP runtime error";
    case SC_ERCCL: return "collective error";
    case SC_ENOHYP: return "no hypothesis: the compatibility graph has no triangle with an inlier";
    case SC_ETOOMANY: return "too many triangles for the workspace cap";
    case SC_ERETRY: return "a candidate blob was too small: repeat with sc_params.shard_cand_level + 1";
    case SC_EBOUND: return "the estimated pruning bound was too high: repeat without SC_FLAG_EST_BOUND";
    default: return "unknown status";
  }
}

void sc_default_params(sc_params* p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->size = sizeof(sc_params);
  p->sigma = 0.1f; p->t_cmp = 0.9f; p->tau = 0.1f; p->min_len = 0.1f;
  p->max_triangles = 50000;
  p->rank_mode = SC_RANK_WEIGHT;
  p->layout = SC_AOS;
  p->shard_rank = 0; p->shard_world = 1; p->shard_block = 1024;
  p->flags = 0; p->max_workspace = 0;
  p->score_mode = SC_SCORE_COUNT; p->shard_cand_level = 0;
}

int sc_create(int device, sc_ctx** out) {
  if (!out) return SC_EINVAL;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) { (void)hipGetLastError(); return SC_EHIP; }  // no GPU: fail loudly
  if (device < 0 || device >= count) return SC_EINVAL;
  sc_ctx* c = new (std::nothrow) sc_ctx();
  if (!c) return SC_ENOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return SC_EHIP;
  }
  c->stream = c->own_stream;
  for (int i = 0; i < N_EVENTS; i++)
    if (hipEventCreateWithFlags(&c->ev[i], hipEventDisableSystemFence) != hipSuccess) { sc_destroy(c); return SC_EHIP; }
  if (hipHostMalloc((void**)&c->pinned, N_PINNED * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
    sc_destroy(c);
    return SC_EHIP;
  }
  *out = c;
  return SC_OK;
}

void sc_destroy(sc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);  // nullptr = the default stream
  Buf* bufs[] = {&c->in_src, &c->in_tgt, &c->planes, &c->S, &c->bits, &c->deg, &c->degp, &c->wpre, &c->ebase, &c->edge_off, &c->scan_tmp,
                 &c->ei, &c->ej, &c->es, &c->ebi, &c->ebj, &c->tcnt, &c->toff, &c->wkey, &c->kcol, &c->ctl, &c->events, &c->blk_gt, &c->blk_eq, &c->blk_minmax, &c->bits2, &c->off_gt,
                 &c->off_eq, &c->sel_ord, &c->sel_key, &c->sortkey, &c->sorted, &c->sort_tmp, &c->tri, &c->tri_rk, &c->key_rk, &c->rt,
                 &c->rt_aos, &c->partial, &c->cnt, &c->key, &c->rt12, &c->mask, &c->refine_tmp, &c->amx_pairs, &c->strong, &c->rowcost, &c->cost_pre, &c->lb_state, &c->lb_ticket, &c->fx_tile, &c->fx_state, &c->fx_mx, &c->fx_part, &c->fx_coef, &c->guard_tmp, &c->fx_frame, &c->ref_cand};
  for (sc_ctx::WgMap& m : c->wg_maps) if (m.buf.p) (void)hipFree(m.buf.p);
  for (Buf* b : bufs) if (b->p) (void)hipFree(b->p);
  for (int i = 0; i < N_EVENTS; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->h_in) (void)hipHostFree(c->h_in);
  if (c->h_out) (void)hipHostFree(c->h_out);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int sc_set_stream(sc_ctx* c, void* hip_stream) {
  if (!c) return SC_EINVAL;
  { const int brc = busy(c); if (brc) return brc; }
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->ev_overhead_us = -1.f;  // event cost differs between streams: calibrate again on the next timed call
  if (hip_stream == SC_STREAM_DEFAULT) c->stream = nullptr;  // the null stream
  else c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return SC_OK;
}

const char* sc_last_error(const sc_ctx* c) { return c ? c->last_error.c_str() : "null context"; }

int sc_set_debug(sc_ctx* c, const sc_debug* d) {
  if (!c) return SC_EINVAL;
  { const int brc = busy(c); if (brc) return brc; }
  if (!d) { c->tn = Tuning(); c->fast_ok = false; c->est_failed = false; c->est_holdoff = 0; c->est_failures = 0; return SC_OK; }
  if (d->size != sizeof(sc_debug)) return SC_EINVAL;
  auto tg_ok = [](uint32_t t) { return t == 0 || t == 4 || t == 8 || t == 16 || t == 32 || t == 64; };
  for (int k = 0; k < 4; k++) if (!tg_ok(d->lanes_per_edge[k])) return SC_EINVAL;
  if (d->lanes_per_edge[0] == 64) return SC_EINVAL;  // the plain counting kernel has no 64-lane form
  if (d->score_split > 256 || (d->compat_rows != 0 && d->compat_rows != 16 && d->compat_rows != 32 && d->compat_rows != 64)) return SC_EINVAL;
  if (d->event_cap != 0 && d->event_cap < 256) return SC_EINVAL;
  Tuning t;
  t.no_events = d->no_events != 0;
  t.event_cap = d->event_cap;
  if (d->compact_self_max >= 0) t.compact_self_max = (size_t)d->compact_self_max;
  if (d->scan_self_max >= 0) t.scan_self_max = (size_t)d->scan_self_max;
  t.cnt_blocks = d->grid_blocks[0]; t.keys_blocks = d->grid_blocks[1]; t.sel_blocks = d->grid_blocks[2]; t.sample_blocks = d->grid_blocks[3];
  if (d->lanes_per_edge[0]) t.tg_count = (int)d->lanes_per_edge[0];
  if (d->lanes_per_edge[1]) t.tg_keys = (int)d->lanes_per_edge[1];
  if (d->lanes_per_edge[2]) t.tg_sample = (int)d->lanes_per_edge[2];
  if (d->lanes_per_edge[3]) t.tg_events = (int)d->lanes_per_edge[3];
  t.sample_edges = d->sample_edges;
  t.score_split = d->score_split;
  t.compat_one_phase = d->compat_one_phase != 0;
  t.compat_rows = (int)d->compat_rows;  // 0 (by size), 16, 32, 64: checked above
  t.compat_store_mode = d->compat_store_mode & 7u;
  t.compat_linear_order = d->compat_linear_order != 0;
  t.sample_mode = d->sample_mode <= 2 ? d->sample_mode : 0u;  // (!= 0 also keeps the estimate off: a certifying form was asked for)
  t.rows_unfused = d->rows_unfused != 0;
  t.score_filter = d->score_filter <= 3 ? d->score_filter : 0u;
  t.filter_splits = d->filter_splits;
  t.filter_queue_cap = d->filter_queue_cap;
  t.filter_lds_queue = d->filter_lds_queue;
  t.filter_blind = d->filter_blind != 0;
  t.gram_kappa_q4 = d->gram_kappa_q4;
  t.gram_ref_late = d->gram_ref_late != 0;
  t.no_fast = d->no_fast != 0;
  t.gram_guard_fail = d->gram_guard_fail != 0;
  t.no_estimate = d->no_estimate != 0;
  t.no_edge_build = d->no_edge_build != 0;
#ifdef SC_ABLATIONS
  t.filter_variant = d->filter_variant;  // (lab builds only: most values are timing-only bodies that return WRONG counts)
#endif
  t.est_margin_pct = d->est_margin_pct;
  c->tn = t;
  c->fast_ok = false;  // (the next call waits: its launch geometry may differ from the last call's)
  c->est_failed = false; c->est_holdoff = 0; c->est_failures = 0;
  return SC_OK;
}

int sc_debug_last(sc_ctx* c, sc_debug_info* out) {
  if (!c || !out || out->size != sizeof(sc_debug_info)) return SC_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  out->c2_kernel = c->filter_on ? (uint32_t)c->filter_mode : 0u;
  out->filter_splits = c->filter_on ? c->fx_plan.splits : 0u;
  out->filter_undecided = 0; out->filter_recounts = 0;
  out->fast_path = (uint32_t)c->fast_state;
  out->gram_guard = c->tn.gram_guard_fail && c->gram_guard != 0 ? 2u : (uint32_t)c->gram_guard; out->gram_guard_worst = c->gram_guard_worst;
  out->prune_bound = c->est_failed_call ? 2u : (uint32_t)c->est_state; out->reserved2 = 0;
  out->gram_near_corr = out->gram_near_hyp = out->gram_rows = out->gram_ref_votes_q8 = 0u; out->gram_ref = 0xFFFFFFFFu;
  out->us_c2_filter = (c->hot_ext && c->filter_on) ? ev_us_raw(c, 4, 11) : 0.f;
  out->n_frames = c->n_frames; out->n_fast_ok = c->n_fast_ok; out->n_fast_repeat = c->n_fast_repeat;
  out->n_est_ok = c->n_est_ok; out->n_est_fail = c->n_est_fail;
  out->cover_edges = c->E_cov; out->cover_triangles = c->M_cov; out->n_hostfree_grow = c->n_spec_grow;
  if (c->filter_on && c->fx_state.p)
    HIPCHK(c, filter_read_counters(c->fx_state.p, c->fx_plan, c->stream, &out->filter_undecided, &out->filter_recounts));
  if (c->filter_on && c->filter_mode == 2 && c->fx_frame.p) {
    uint32_t fr[5];
    HIPCHK(c, filter_read_frame(c->fx_frame.p, c->stream, fr));
    out->gram_near_corr = fr[0]; out->gram_near_hyp = fr[1]; out->gram_rows = fr[2]; out->gram_ref = fr[3]; out->gram_ref_votes_q8 = fr[4];
  }
  return SC_OK;
}

}  // extern "C" (the two halves below are internal)

namespace {

// phase 1, first half: staging, stage A, the edge list and this process's share of the pruning sample
int hyp_begin(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p, uint32_t* d_hist,
              uint32_t part, uint32_t parts) {
  int rc = check_params(p);
  if (rc) return rc;
  if (c->pending) {  // (every entry that starts a call on the context comes through here or through sc_shard_compat_device)
    c->last_error = "a call is outstanding on this context (sc_wait first)";
    return SC_EINVAL;
  }
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false;
  c->begun = false;
  c->timed_trikeys = false;
  c->regular = false;
  if (!c->spec_on) c->E_cov = c->M_cov = 0;  // (only the entry that has just planned a host-free call leaves spec_on set)
  if (!c->est_allowed) c->est_failed_call = false;  // (an entry point that never estimates)
  if ((rc = set_timing(c, p))) return rc;
  c->refine = (p->flags & SC_FLAG_REFINE) != 0;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  c->dv = derive(p);
  c->params = *p;
  // SC_FLAG_EST_BOUND outside the sc_shard_* phase API: sc_hypothesize_device only (stages A and B replicated, every rank takes the
  // whole estimating sample; a shared histogram — sc_hypothesize_begin / _end_device — is a certifying sample's)
  if ((p->flags & SC_FLAG_EST_BOUND) && !(d_hist == nullptr && parts == 1 && c->est_allowed)) {
    c->last_error = "SC_FLAG_EST_BOUND belongs to the sc_shard_* phase API and to sc_hypothesize_device";
    return SC_EINVAL;
  }
  c->build = d_hist == nullptr && parts == 1 && edge_build_ok(c, p, n);
  if ((rc = rec(c, 0))) return rc;
  if ((rc = stage_inputs(c, d_src, d_tgt, n, p))) return rc;
  if ((rc = rec(c, 1))) return rc;
  if ((rc = run_compat(c, !(p->flags & SC_FLAG_NO_DENSE_S)))) return rc;
  if ((rc = rec(c, 2))) return rc;
  if ((rc = run_row_stats(c, may_prune(p), true))) return rc;
  if ((rc = run_edges(c, p, d_hist, part, parts))) return rc;
  c->begun = true;
  return SC_OK;
}

int run_stage_c(sc_ctx* c, uint64_t* d_key, sc_stats* stats);

// C2: the counts of this shard's hypotheses into c->partial; *rows: rows of c->partial the arg-max has to add up.
// SC_SCORE_COUNT runs the matrix-pipe filter + the exact fix-up (sc_score.hip); the truncated scores, and
// sc_debug.score_filter = 1, the plain fp32 kernel.
// C2 by filter?  By mode / size / knobs (score_uses_filter), and — unless forced — only when tau is on a scale the filter
// can bound (a tau below ~2.4e-4 of the scene's extent, or above 8 x, would send every wave to the exact recount, which
// is slower than the plain kernel): the staging kernel has told the host the coordinate maxima long before this point.
// Evaluated ONCE per call (decide_filter stores the answer in the context): the word it reads is written by the GPU, so
// two evaluations could disagree.  On the path the maxima have always arrived by then (the host has polled results of
// later kernels of the same stream); the stage hook waits for the word.  Not known (sc_debug.filter_blind forces that
// case) means "assume in range": the filter itself hands whatever it cannot bound to the exact recount, so the counts
// are the same either way, only slower.
int use_filter(const sc_ctx* c, const sc_params* p, const Shard& sh) {
  const bool blind = c->tn.filter_blind;
  // flag, THEN data: the staging kernel writes the six box words before the word of maxima, so the maxima are loaded with
  // acquire semantics and the boxes only afterwards (the fence used to sit behind both loads, which only the
  // load order of x86 made right)
  uint64_t mx = blind ? ~0ull : __atomic_load_n(&c->pinned[13], __ATOMIC_ACQUIRE);
  uint64_t box[6];
  for (int k = 0; k < 6; k++) box[k] = *const_cast<volatile uint64_t*>(&c->pinned[16 + k]);
  // A host-free call enqueued while the previous frame still runs (sc_register_device_async on a second context of the stream)
  // gets here before its own staging kernel has started: it is a repetition of the last call's shape, so it decides by the last
  // call's words.  Any choice gives the same counts (see above); decided blind, every such call ran the linear filter — at C2 53 us
  // against 27 (tools/r4/b2b.py).
  if (!blind && mx == ~0ull && c->spec_on && c->mx_last != ~0ull) {
    mx = c->mx_last;
    for (int k = 0; k < 6; k++) box[k] = c->box_last[k];
  }
  return score_filter_mode(p->score_mode, c->tn, c->n, sh.ld_local, mx, (blind || mx == ~0ull) ? nullptr : box, c->dv.tau2);
}
// The Gram filter's bound rests on a measured model of the matrix pipe (sc_score.hip, gram_guard_kernel): probe the pipe this
// context runs on, once, the first time a call would choose that filter.  0 not run, 1 the model holds, 2 violated.
int run_gram_guard(sc_ctx* c) {
  if (c->gram_guard != 0) return SC_OK;
  ENSURE(c, c->guard_tmp, 64);
  float worst = 0.f; bool sub_ok = false; uint32_t compared = 0;
  const hipError_t e = gram_guard_probe(c->guard_tmp.p, c->stream, &worst, &sub_ok, &compared);
  if (e != hipSuccess) return fail_hip(c, e, "gram_guard_probe");
  c->gram_guard_worst = worst;
  c->gram_guard = (sub_ok && compared != 0 && worst <= gram_guard_limit()) ? 1 : 2;
  return SC_OK;
}
int decide_filter(sc_ctx* c, const sc_params* p, const Shard& sh) {
  c->filter_mode = use_filter(c, p, sh);
  if (c->filter_mode == 2) {
    const int rc = run_gram_guard(c);
    if (rc) return rc;
    if (c->gram_guard == 2 || c->tn.gram_guard_fail) {
      // this pipe does not add the way the Gram bound assumes (or a test says so): the linear filter, whose bound allows a
      // truncating accumulation, where tau is on its scale; the plain kernel otherwise
      const uint64_t mx = c->tn.filter_blind ? ~0ull : __atomic_load_n(&c->pinned[13], __ATOMIC_ACQUIRE);
      c->filter_mode = filter_in_range(mx, c->dv.tau2) ? 1 : 0;
    }
  }
  c->filter_on = c->filter_mode != 0;
  if (c->filter_on) c->fx_plan = filter_plan(c->n, sh.ld_local, c->tn, (uint32_t)c->filter_mode);
  return SC_OK;
}

// the filter's buffers for this shard, and the job that fills the tile / clears the state
int filter_job(sc_ctx* c, const Shard& sh, FilterTileJob* job) {
  const FilterPlan& fp = c->fx_plan;
  ENSURE(c, c->fx_tile, fp.tile_bytes);
  ENSURE(c, c->fx_state, fp.state_bytes);
  if (fp.coef_bytes) ENSURE(c, c->fx_coef, fp.coef_bytes);
  uint32_t* mx = c->fx_mx.as<uint32_t>();
  if (fp.mode == 2) { const int frc = ensure_frame(c); if (frc) return frc; }
  *job = filter_tile_job(fp, mx, nullptr, c->fx_tile.p, c->fx_state.p, c->fx_coef.p, sh.ld_local, c->dv.tau2, c->fx_frame.p);
  return SC_OK;
}

int run_score(sc_ctx* c, const sc_params* p, const Shard& sh, uint32_t* rows, bool tile_done, hipEvent_t ev0 = nullptr,
              hipEvent_t ev1 = nullptr, hipEvent_t ev_mid = nullptr) {
  if (c->filter_on) {  // decide_filter() ran earlier in this call
    const FilterPlan& fp = c->fx_plan;
    *rows = fp.splits;
    ENSURE(c, c->partial, (size_t)fp.splits * sh.ld_local * 4);
    if (!tile_done) {  // (the stage hook; the path builds the tile inside the Kabsch launch)
      FilterTileJob job;
      int rc = filter_job(c, sh, &job);
      if (rc) return rc;
      if (fp.mode == 2)  // the frame first: tile and coefficients are made in it (the hypotheses exist already: it votes among them)
        launch_gram_ref(points_of(c), TriSource{}, sh, c->rt.as<float>(), nullptr, c->fx_mx.as<uint32_t>(), c->dv.tau2, c->tn,
                        c->fx_frame.p, c->stream);
      launch_filter_tile(points_of(c), job, c->stream);
      if (fp.mode == 2) launch_gram_coef(c->rt.as<float>(), sh, c->dv.tau2, job.coef, c->stream);
    }
    launch_score_filter(points_of(c), c->rt.as<float>(), c->rt_aos.as<float>(), sh, c->dv, fp, c->fx_tile.p, c->fx_state.p,
                        c->fx_coef.p, c->fx_frame.p, c->partial.as<uint32_t>(), c->tn, c->stream, ev0, ev1, ev_mid);
    return SC_OK;
  }
  *rows = score_chunks(c->n, sh.ld_local);
  if (sh.ld_local) ENSURE(c, c->partial, (size_t)*rows * sh.ld_local * 4);
  launch_score(points_of(c), c->rt.as<float>(), c->rt_aos.as<float>(), sh, c->dv, p->score_mode, c->partial.as<uint32_t>(),
               c->tn, c->stream, ev0, ev1);
  return SC_OK;
}

// phase 1, second half: prune with the (summed) histogram, enumerate, select, then stage C on this rank's share
int hyp_end(sc_ctx* c, const uint32_t* d_hist, uint64_t* d_key, sc_stats* stats) {
  const sc_params* p = &c->params;
  int rc;
  c->begun = false;
  if ((rc = run_select(c, p, d_hist, false))) return rc;
  return run_stage_c(c, d_key, stats);
}

// stage C on this rank's blocks of the (replicated) selection: C1, C2, arg-max key pair -> d_key
int run_stage_c(sc_ctx* c, uint64_t* d_key, sc_stats* stats) {
  const sc_params* p = &c->params;
  int rc;
  if ((rc = rec(c, 3))) return rc;
  Shard sh;
  sh.T_eff = c->T_eff;
  sh.block = p->shard_block ? p->shard_block : 1024u;
  sh.rank = (uint32_t)p->shard_rank;
  sh.world = (uint32_t)p->shard_world;
  sh.n_local = shard_local_count(sh.T_eff, sh.block, sh.rank, sh.world);
  sh.ld_local = (uint32_t)(((uint64_t)sh.n_local + 255u) / 256u * 256u);  // T <= 2^32 - 256 (check_params)
  c->sh = sh;
  if (sh.n_local) {
    ENSURE(c, c->rt, (size_t)12 * sh.ld_local * 4);
    ENSURE(c, c->cnt, (size_t)sh.ld_local * 4);
    if ((rc = decide_filter(c, p, sh))) return rc;
    const bool filter = c->filter_on;
    const bool aos = filter;  // (the exact pass reads 12 consecutive floats per hypothesis)
    if (aos) ENSURE(c, c->rt_aos, (size_t)12 * sh.ld_local * 4);
    FilterTileJob job;
    if (filter && (rc = filter_job(c, sh, &job))) return rc;
    // host-free call: the selection may turn out shorter than the T this launch was sized for (then the call is repeated);
    // its real length is what the select left in sel.want
    const uint64_t* t_eff_dev = (c->spec_on || c->est_active) ? &c->ctl.as<ControlBlock>()->sel.want : nullptr;
    if (filter && c->fx_plan.mode == 2 && !c->ref_done)  // the Gram filter's reference frame, where the counting pass did not carry the vote: among 64 hypotheses of the selection
      launch_gram_ref(points_of(c), tri_source_of(c), sh, nullptr, t_eff_dev, c->fx_mx.as<uint32_t>(), c->dv.tau2, c->tn,
                      c->fx_frame.p, c->stream);
    launch_kabsch(points_of(c), tri_source_of(c), sh, c->rt.as<float>(), aos ? c->rt_aos.as<float>() : nullptr,
                  filter ? &job : nullptr, c->stream, t_eff_dev);
  } else {
    c->filter_on = false; c->filter_mode = 0;
  }
  // SC_FLAG_TIMING_HOT: the score stage's duration from the dispatch packets of its own kernels (no record on the stream:
  // the two records of a bracket opened ~4.5 us gaps before and after the stage — 3 % of a C2 step)
  const bool hot_ext = c->timing_hot && sh.n_local != 0;
  c->hot_ext = hot_ext;
  if (!hot_ext && (rc = rec(c, 4))) return rc;
  uint32_t score_rows = 0;
  if ((rc = run_score(c, p, sh, &score_rows, true, hot_ext ? c->ev[4] : nullptr, hot_ext ? c->ev[5] : nullptr, hot_ext ? c->ev[11] : nullptr))) return rc;
  if (!hot_ext && (rc = rec(c, 5))) return rc;
  ENSURE(c, c->amx_pairs, argmax_scratch_bytes(sh.ld_local));
  // the winner pair: reduced by the launch for a caller that gathers it; left as one pair per workgroup for this context's own
  // finalize step (sc_register*: d_key is the context's own word), whose workgroups reduce them themselves
  const bool own_pairs = d_key == c->key.as<uint64_t>() && sh.n_local != 0;
  c->amx_blocks = own_pairs ? argmax_blocks(sh.ld_local) : 0u;
  launch_argmax(sh, c->partial.as<uint32_t>(), score_rows, c->T_eff ? c->sel_key.as<uint32_t>() : nullptr,
                c->cnt.as<uint32_t>(), c->amx_pairs.as<uint64_t>(), &c->ctl.as<ControlBlock>()->amx_ticket, own_pairs ? nullptr : d_key, c->stream);
  if ((rc = rec(c, 6))) return rc;
  HIPCHK(c, hipGetLastError());
  c->have_hyp = true;
  fill_stats(c, stats);  // counts only: the event times are read by sc_finalize_device, after ITS synchronisation
  return SC_OK;
}

}  // namespace

extern "C" {

int sc_hypothesize_device(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p,
                          uint64_t* d_key, sc_stats* stats) {
  if (!c || !d_src || !d_tgt || !d_key) return SC_EINVAL;
  // SC_FLAG_EST_BOUND (r04b): stage B pruned by an estimated bound, like sc_register_device's — the select verifies it and
  // sc_finalize(_gathered)_device returns SC_EBOUND when it was too high (on every rank of a job alike: stages A and B are
  // replicated and deterministic): the caller repeats both calls without the flag
  const bool est = p && p->size == sizeof(sc_params) && (p->flags & SC_FLAG_EST_BOUND) != 0;
  if (c->pending) { c->last_error = "a call is outstanding on this context (sc_wait first)"; return SC_EINVAL; }
  // r04c: with the flag, a call that repeats the last call's shape on this context is enqueued HOST-FREE, as sc_register_device's
  // is (fast_plan): no wait for stage B's two counts — the rank's host thread is free to enqueue the job's next frame.  The
  // finalize call validates; a count that outgrew what the launches covered comes back as SC_EBOUND too ("repeat both calls
  // without the flag" — the waiting, certifying way handles every case), on every rank alike: stages A and B are replicated.
  c->spec_on = false;
  c->fast_state = 0;
  if (est && check_params(p) == SC_OK && fast_plan(c, n, p)) c->spec_on = true;
  c->est_allowed = est;
  if (est) c->est_failed = false;  // (the CALLER decides when to stop estimating on this entry)
  int rc = hyp_begin(c, d_src, d_tgt, n, p, nullptr, 0, 1);  // the whole sample, into the context's histogram
  if (rc == SC_OK) rc = hyp_end(c, nullptr, d_key, stats);
  c->est_allowed = false;
  if (rc != SC_OK && c->spec_on) { c->spec_on = false; c->fast_ok = false; }
  return rc;
}

int sc_hypothesize_begin_device(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p,
                                uint32_t* d_hist, sc_stats* stats) {
  if (!c || !d_src || !d_tgt || !d_hist) return SC_EINVAL;
  if (!p || p->size != sizeof(sc_params)) return SC_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = hyp_begin(c, d_src, d_tgt, n, p, d_hist, (uint32_t)p->shard_rank, (uint32_t)p->shard_world);
  if (rc == SC_OK) fill_stats(c, stats);
  return rc;
}

int sc_hypothesize_end_device(sc_ctx* c, const uint32_t* d_hist, uint64_t* d_key, sc_stats* stats) {
  if (!c || !d_hist || !d_key) return SC_EINVAL;
  { const int brc = busy(c); if (brc) return brc; }
  if (!c->begun) { c->last_error = "sc_hypothesize_end_device without a preceding sc_hypothesize_begin_device"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  return hyp_end(c, d_hist, d_key, stats);
}

// ---- sharded A + B ------------------------------------------------------------------------

int sc_shard_plan_query(const sc_params* p, int64_t n, sc_shard_plan* out) {
  if (!out || out->size != sizeof(sc_shard_plan)) return SC_EINVAL;
  const int rc = check_params(p);
  if (rc) return rc;
  if (n < 3 || n > (1 << 24) || p->shard_world > 64) return SC_EINVAL;
  const uint64_t ld = (uint64_t)round_up((int)n, 64), W = ld >> 6, G = (uint64_t)p->shard_world;
  const uint64_t R = (((uint64_t)n + G - 1) / G + 63) / 64 * 64;  // rows per rank, a multiple of the tile block
  out->rows_per_rank = (uint32_t)R;
  out->words_per_row = (uint32_t)W;
  out->bits_bytes_per_rank = R * W * 8;
  out->bits_bytes_total = G * R * W * 8;
  out->cand_bytes_per_rank = cand_blob_bytes(cand_cap(p->max_triangles, (uint32_t)p->shard_world, p->shard_cand_level));
  return SC_OK;
}

int sc_shard_compat_device(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p,
                           void* d_bits_all) {
  if (!c || !d_src || !d_tgt || !d_bits_all) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  if (p->shard_world > 64) return SC_EINVAL;
  if (c->pending) { c->last_error = "a call is outstanding on this context (sc_wait first)"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false; c->begun = false; c->timed_trikeys = false;
  c->spec_on = false;
  if ((rc = set_timing(c, p))) return rc;
  c->refine = (p->flags & SC_FLAG_REFINE) != 0;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  c->dv = derive(p);
  c->params = *p;
  c->shard_phase = 0; c->cand_all = nullptr;
  c->build = false;
  if ((rc = rec(c, 0))) return rc;
  if ((rc = stage_inputs(c, d_src, d_tgt, n, p))) return rc;
  if ((rc = rec(c, 1))) return rc;
  sc_shard_plan plan; plan.size = sizeof plan;
  if ((rc = sc_shard_plan_query(p, n, &plan))) return rc;
  const int R = (int)plan.rows_per_rank;
  const int r0 = p->shard_rank * R < c->n ? p->shard_rank * R : c->n, r1 = r0 + R < c->n ? r0 + R : c->n;
  const size_t W = (size_t)c->ld >> 6;
  const bool dense = !(p->flags & SC_FLAG_NO_DENSE_S);
  if (dense && r1 > r0) ENSURE(c, c->S, (size_t)(r1 - r0) * c->ld * sizeof(float));  // this rank's rows of S only
  ENSURE(c, c->deg, (size_t)c->n * sizeof(uint32_t));
  ENSURE(c, c->degp, (size_t)c->n * sizeof(uint32_t));
  ENSURE(c, c->wpre, (size_t)c->n * W * sizeof(uint32_t));
  c->bits_cur = static_cast<uint64_t*>(d_bits_all);
  c->sharded_ab = true;
  // world == 1: [0, n) -> the symmetric tiles; else the one-sided row-block form, bit rows at their global index
  launch_compat(points_of(c), c->dv, dense ? c->S.as<float>() : nullptr, c->bits_cur, r0, r1, c->tn, c->stream);
  if ((rc = rec(c, 2))) return rc;
  HIPCHK(c, hipGetLastError());
  c->shard_phase = 1;
  return SC_OK;
}

int sc_shard_edges_device(sc_ctx* c, uint32_t* d_hist) {
  if (!c || !d_hist) return SC_EINVAL;
  { const int brc = busy(c); if (brc) return brc; }
  if (!c->sharded_ab || c->shard_phase != 1) { c->last_error = "sc_shard_edges_device: call sc_shard_compat_device first"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const sc_params* p = &c->params;
  c->shard_phase = 0;
  int rc;
  if ((rc = run_row_stats(c, may_prune(p), true))) return rc;
  if ((rc = run_edges(c, p, d_hist, (uint32_t)p->shard_rank, (uint32_t)p->shard_world))) return rc;
  c->shard_phase = 2;
  return SC_OK;
}

int sc_shard_select_device(sc_ctx* c, const uint32_t* d_hist, void* d_cand_mine) {
  if (!c || !d_hist || !d_cand_mine) return SC_EINVAL;
  { const int brc = busy(c); if (brc) return brc; }
  if (!c->sharded_ab || c->shard_phase != 2) { c->last_error = "sc_shard_select_device: call sc_shard_edges_device first"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const sc_params* p = &c->params;
  c->shard_phase = 0;
  int rc;
  if ((rc = run_select(c, p, d_hist, false))) return rc;   // this rank's own top-T, in (i,j,k) order
  const CandBlob b = cand_blob(d_cand_mine, cand_cap(p->max_triangles, (uint32_t)p->shard_world, p->shard_cand_level));
  if (c->M == 0) {  // nothing enumerated here (also E == 0): an empty blob
    HIPCHK(c, hipMemsetAsync(b.hdr, 0, CAND_HDR_WORDS * 8, c->stream));
  } else {
    launch_cand_emit(c->sel_ord.as<uint64_t>(), c->sel_key.as<uint32_t>(), c->kcol.as<uint2>(), c->ei.as<uint32_t>(),
                     c->ej.as<uint32_t>(), &c->ctl.as<ControlBlock>()->sel, c->toff.as<uint64_t>(), c->E, c->T_eff,
                     // a full-length list (T entries) is never "cut": the global top-T takes at most T from one rank
                     select_want(c, p) < p->max_triangles ? (uint64_t)c->T_eff : ~0ull, b, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  c->shard_phase = 3;
  return SC_OK;
}

int sc_shard_score_device(sc_ctx* c, const void* d_cand_all, uint64_t* d_key, sc_stats* stats) {
  if (!c || !d_cand_all || !d_key) return SC_EINVAL;
  { const int brc = busy(c); if (brc) return brc; }
  if (!c->sharded_ab || c->shard_phase != 3) { c->last_error = "sc_shard_score_device: call sc_shard_select_device first"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  const sc_params* p = &c->params;
  c->shard_phase = 0;
  hipStream_t st = c->stream;
  const uint32_t T = p->max_triangles, G = (uint32_t)p->shard_world;
  const size_t cap = cand_cap(T, G, p->shard_cand_level);
  const size_t blob_bytes = cand_blob_bytes(cap);
  c->cand_all = d_cand_all; c->cand_bytes = blob_bytes;
  // Merge: the same exact select + (i,j,k)-order compaction as on one GPU, over the concatenated candidate keys.
  ControlBlock* ctl = c->ctl.as<ControlBlock>();
  SelectState* sel = &ctl->sel;
  const bool window_known = p->rank_mode == SC_RANK_WEIGHT && 3.0f * p->t_cmp * 0.999f >= 2.0f;
  const KeyView view = cand_view(d_cand_all, blob_bytes, G, cap);
  const size_t nb = compact_blocks(view.M);
  ENSURE(c, c->blk_gt, nb * 4);
  ENSURE(c, c->blk_eq, nb * 4);
  ENSURE(c, c->off_gt, (nb + 1) * 8);
  ENSURE(c, c->off_eq, (nb + 1) * 8);
  ENSURE(c, c->scan_tmp, scan_temp_bytes(nb));
  ENSURE(c, c->sel_ord, (size_t)T * 8);
  ENSURE(c, c->sel_key, (size_t)T * 4);
  arm_word(c, 6);
  c->pinned[15] = 0;  // "an estimated bound promised T keys above it and the merged candidates hold fewer" (merge_prepare_kernel)
  launch_merge_prepare(d_cand_all, blob_bytes, G, T, window_known, &ctl->klb, sel, &c->pinned[6], st,
                       c->est_active ? &c->pinned[15] : nullptr);
  launch_select_rounds(view, sel, window_known ? 2 : 3, c->tn, st);
  c->pinned[12] = 0;  // "a cut candidate list could have mattered": read by the finalize call (SC_ERETRY)
  { const int crc = run_compaction(c, view, nb, window_known ? 2 : 3, nullptr); if (crc) return crc; }
  launch_merge_check(d_cand_all, blob_bytes, G, sel, &c->pinned[12], st);  // (behind the compaction: its counting kernel is what resolves the last round into k*)
  // the merged length, published by merge_prepare long before the compaction ends: the poll costs no GPU time
  { const int wrc = wait_word(c, 6); if (wrc) return wrc; }
  c->T_eff = (uint32_t)c->pinned[6];
  c->M = c->M_total = c->pinned[7];
  return run_stage_c(c, d_key, stats);
}

int sc_finalize_device(sc_ctx* c, const uint64_t* d_key, float* d_Rt, uint8_t* d_mask, sc_stats* stats) {
  return sc_finalize_gathered_device(c, d_key, 1, d_Rt, d_mask, stats);
}

}  // extern "C" (the halves of the finalize step and the host-free machinery are internal)

namespace {

constexpr int SC_ESPEC = -100;  // internal: a host-free call failed validation (never leaves the library)

// phase 2, first half: the winner / mask kernel (and the optional refit) are enqueued; nothing is waited for
int finalize_enqueue(sc_ctx* c, const uint64_t* d_keys, int n_pairs, float* d_Rt, uint8_t* d_mask) {
  int rc;
  if ((rc = rec(c, 7))) return rc;
  ControlBlock* ctl = c->ctl.as<ControlBlock>();
  arm_word(c, 8);
  // a host-free call's kernels have published nothing yet: this one hands the counts and the coordinate statistics over with the winner
  DeferredPub dp{c->edge_off.as<uint64_t>() + c->n, c->toff.as<uint64_t>() + c->E, c->fx_mx.as<uint32_t>(),
                 reinterpret_cast<unsigned long long*>(c->pinned), (c->n_fast_ok & 63u) == 63u ? 1 : 0};
  if (d_keys == c->key.as<uint64_t>() && n_pairs == 1 && c->amx_blocks) { d_keys = c->amx_pairs.as<uint64_t>(); n_pairs = (int)c->amx_blocks; }
  launch_finalize(points_of(c), tri_source_of(c), c->sh, c->sh.n_local ? c->rt.as<float>() : nullptr,
                  c->T_eff ? c->sel_key.as<uint32_t>() : nullptr, c->T_eff, d_keys, n_pairs,
                  ctl->key2, c->dv.tau2, d_Rt, d_mask, &ctl->fin_word, &c->pinned[8], c->stream,
                  c->spec_on ? &dp : nullptr);
  if (c->refine) {  // fp64 least-squares refit over the winner's inliers (mask unchanged)
    ENSURE(c, c->refine_tmp, refine_scratch_bytes(c->n));
    launch_refine(points_of(c), d_mask, ctl->key2, c->refine_tmp.as<double>(), d_Rt, c->stream);
  }
  return rec(c, 8);
}

// the parameters that make two calls "the same shape" (a host-free call repeats the last call's; timing flags aside)
bool same_shape(const sc_params* p, const sc_params* q) {
  constexpr uint32_t TIMING_BITS = SC_FLAG_TIMING | SC_FLAG_TIMING_HOT | SC_FLAG_TIMING_ONE | (15u << 8);
  return p->sigma == q->sigma && p->t_cmp == q->t_cmp && p->tau == q->tau && p->min_len == q->min_len &&
         p->max_triangles == q->max_triangles && p->rank_mode == q->rank_mode && p->layout == q->layout &&
         p->shard_world == q->shard_world && p->shard_rank == q->shard_rank && p->shard_block == q->shard_block &&
         p->score_mode == q->score_mode && (p->flags & ~TIMING_BITS) == (q->flags & ~TIMING_BITS);
}

// what the next call on this context may assume (fast_plan): the call that just completed was a "regular" one
void note_completed(sc_ctx* c, bool regular) {
  c->fast_ok = regular && !c->sharded_ab && c->E >= 4096 && c->T_eff == c->params.max_triangles;  // (stages A and B whole on this GPU: one rank, or replicated ranks)
  if (c->n != c->last_n || !same_shape(&c->params, &c->last_p)) { c->E_hi[0] = c->E_hi[1] = c->M_hi[0] = c->M_hi[1] = 0; c->hi_n = 0; c->hi_seen = 0; }
  if (c->fast_ok) {  // (the counts of a regular call of this shape: what a host-free repetition has to cover)
    if (c->hi_seen < 0xFFFFFFFFu) c->hi_seen++;
    if (c->E > c->E_hi[0]) c->E_hi[0] = c->E;
    if (c->M > c->M_hi[0]) c->M_hi[0] = c->M;
    if (++c->hi_n >= sc_ctx::HI_WINDOW) { c->E_hi[1] = c->E_hi[0]; c->M_hi[1] = c->M_hi[0]; c->E_hi[0] = c->M_hi[0] = 0; c->hi_n = 0; }
  }
  c->E_last = c->E; c->M_last = c->M; c->last_n = c->n;
  c->last_p = c->params;
  if (c->est_failed && !c->est_failed_call && c->est_holdoff != 0 && --c->est_holdoff == 0) c->est_failed = false;  // (the repeat itself does not count)
  // (the call is complete: its staging kernel's words have arrived — unless it was a host-free call, whose statistics travel with
  // the winner only now and then: DeferredPub; the last ones known stay)
  const uint64_t mx_now = __atomic_load_n(&c->pinned[13], __ATOMIC_ACQUIRE);
  if (mx_now != ~0ull) {
    c->mx_last = mx_now;
    for (int k = 0; k < 6; k++) c->box_last[k] = *const_cast<volatile uint64_t*>(&c->pinned[16 + k]);
  }
  // room for the event list of a call like this one: an event holds >= 1 triangle, so 2 M records can only overflow a region
  // on a fill 2 x off the mean (the regions fill evenly: a wave moves to the next one with every flush)
  if (c->use_events && c->ev_capacity < 2 * c->M) c->ev_capacity = 2 * c->M < (1ull << 28) ? 2 * c->M : (1ull << 28);
}

// phase 2, second half: wait for the winner; a host-free call is validated here
int finalize_wait(sc_ctx* c, sc_stats* stats) {
  int rc;
  // The finalize kernel publishes key / position / rank.  On a caller-provided stream (sc_set_stream) d_Rt and d_mask
  // are complete in stream order, like any other work the caller enqueues there; on the context's private stream —
  // which the caller cannot order against — and when the per-stage events are read below, wait for everything.
  const bool was_spec = c->spec_on;
  c->spec_on = false;  // (on EVERY way out of here: a timed-out wait below must not leave the covers to the next entry — ADVICE r04)
  if (c->timing || c->stream == c->own_stream) HIPCHK(c, hipStreamSynchronize(c->stream));
  else if (c->timing_one == 6) HIPCHK(c, hipEventSynchronize(c->ev[8]));  // the mask bracket ends after the kernel polled below
  if ((rc = wait_word(c, 8))) { c->fast_ok = false; return rc; }
  HIPCHK(c, hipGetLastError());
  if (was_spec) {
    // Host-free call: every kernel of it has finished (the winner word is the last thing the stream writes), so the two
    // counts and the flags are final.  The launches covered E_cov edges and M_cov keys and assumed T triangles exist, an
    // event list that did not overflow and a graph big enough to prune: anything else and the outputs are void — the
    // caller (sc_wait) repeats the call the waiting way, which handles every one of these cases.
    if ((uint32_t)c->pinned[1] != 0) { c->fast_ok = false; c->last_error = "non-finite input coordinate"; return SC_EINVAL; }
    // (the finalize kernel handed both counts over in one word: DeferredPub)
    const uint64_t em = c->pinned[0];
    const uint64_t E = em == PIN_PENDING ? PIN_PENDING : (em & 0xFFFFFFFFull), M = em == PIN_PENDING ? PIN_PENDING : (em >> 32);
    const bool ok = E != PIN_PENDING && M != PIN_PENDING && E >= 4096 && E <= c->E_cov && M <= c->M_cov &&
                    M >= (uint64_t)c->params.max_triangles && (uint32_t)c->pinned[5] == 0;
    if (!ok) {
      char buf[256];  // (why, for sc_last_error: the repeat itself is silent)
      snprintf(buf, sizeof buf, "host-free call repeated: edges %llu (covered %llu), triangles %llu (covered %llu, T %u), event overflow %u",
               (unsigned long long)E, (unsigned long long)c->E_cov, (unsigned long long)M, (unsigned long long)c->M_cov,
               c->params.max_triangles, (uint32_t)c->pinned[5]);
      c->last_error = buf;
      c->fast_ok = false;
      if ((uint32_t)c->pinned[5] != 0 && c->ev_capacity < (1ull << 28)) c->ev_capacity *= 2;  // (the repeat must not overflow again)
      return SC_ESPEC;
    }
    c->E = E; c->M = c->M_total = M;
    c->fast_state = 1;
    c->regular = true;
  }
  c->est_state = c->est_active ? 1 : 0;
  if (c->sharded_ab && c->est_active && (c->pinned[15] != 0 || c->est_void)) {
    // SC_FLAG_EST_BOUND: every rank holds the same gathered blobs, so every rank ends here together; the caller repeats the call
    // without the flag (include/saccot.h)
    c->last_error = "the estimated pruning bound was too high for this input: repeat the call without SC_FLAG_EST_BOUND";
    return SC_EBOUND;
  }
  if (c->pinned[14] != 0 || (c->est_active && c->est_void)) {
    // select_round_kernel: fewer keys at or above the pruning bound than it promised (or the bound went unverified)
    if (!c->est_active) {  // a CERTIFIED bound holds by construction: this would be a defect, not an input
      c->last_error = "internal: the select found fewer keys above a certified pruning bound than the certificate counted";
      return SC_EHIP;
    }
    if (c->params.flags & SC_FLAG_EST_BOUND) {  // sc_hypothesize_device + sc_finalize*_device: the caller repeats (every rank alike)
      c->last_error = "the estimated pruning bound was too high for this input: repeat the call without SC_FLAG_EST_BOUND";
      c->est_failed_call = true;
      return SC_EBOUND;
    }
    c->last_error = "estimated pruning bound too high: call repeated with a certifying sample";
    c->est_failed = true;  // this context certifies for a while (note_completed counts the hold-off down)
    c->est_failures++;
    c->est_holdoff = 64u << (c->est_failures - 1 < 6 ? c->est_failures - 1 : 6);
    c->est_failed_call = true;
    c->fast_ok = false;
    return SC_ESPEC;
  }
  if (c->sharded_ab && c->cand_all && c->pinned[12] != 0) {
    // merge_check_kernel (an earlier kernel of this stream: its system-scope store is visible once the winner word is):
    // some rank's candidate list was cut at a key the merged threshold does not clear
    c->last_error = "a candidate blob was too small for this input: repeat the call with sc_params.shard_cand_level raised by one";
    return SC_ERETRY;
  }
  if (c->pinned[9] == ~0ull) {  // finalize_kernel: a pair decodes to a position outside the selection (outputs: identity, zero mask)
    c->last_error = "a winner key pair points outside the selected list (stale / uninitialised pair, or ranks that disagree on T or the parameters)";
    return SC_EINVAL;
  }
  note_completed(c, c->regular);
  const uint64_t key = c->pinned[8];
  if (stats && stats->size == sizeof(sc_stats)) {
    fill_stats(c, stats);
    stats->best_count = (uint32_t)(key >> 32);
    stats->best_rank = key ? (uint32_t)(c->pinned[9] >> 32) : 0u;  // (rank index << 32 | position: one word beside the key)
    if (c->timing) {
      stats->us_stage = ev_us(c, 0, 1);
      stats->us_compat = ev_us(c, 1, 2);
      stats->us_triangles = ev_us(c, 2, 3);  // includes its two 8-byte read-backs
      stats->us_trikeys = c->timed_trikeys ? ev_us(c, 9, 10) : 0.f;
      stats->us_kabsch = ev_us(c, 3, 4);
      stats->us_score = ev_us(c, 4, 5);
      stats->us_argmax = ev_us(c, 5, 6);
      stats->us_mask = ev_us(c, 7, 8);
      stats->us_total = stats->us_stage + stats->us_compat + stats->us_triangles + stats->us_kabsch +
                        stats->us_score + stats->us_argmax + stats->us_mask;
    } else if (c->timing_hot) {
      stats->us_score = c->hot_ext ? ev_us_raw(c, 4, 5) : ev_us(c, 4, 5);
    } else if (c->timing_one >= 0) {  // one bracket, recorded on the hot path (speculative launches on)
      float* dst[7] = {&stats->us_stage, &stats->us_compat, &stats->us_triangles, &stats->us_kabsch, &stats->us_score,
                       &stats->us_argmax, &stats->us_mask};
      *dst[c->timing_one] = ev_us(c, STAGE_EV[c->timing_one][0], STAGE_EV[c->timing_one][1]);
    }
  }
  return key ? SC_OK : SC_ENOHYP;
}

// May this call be enqueued host-free?  Only a repetition of the last call's shape on this context, and only when that call
// was regular (note_completed); sets what the launches cover: the last counts plus half, within what the arrays hold.
bool fast_plan(sc_ctx* c, int64_t n, const sc_params* p) {
  if (!c->fast_ok || c->tn.no_fast || c->tn.no_events || n != c->last_n) return false;
  if (p->flags & (SC_FLAG_TIMING | SC_FLAG_EXACT_TOTAL | SC_FLAG_NO_PRUNE)) return false;
  if (!same_shape(p, &c->last_p)) return false;
  if (!(p->rank_mode == SC_RANK_WEIGHT && 3.0f * p->t_cmp * 0.999f >= 2.0f)) return false;  // the a-priori select window
  uint64_t ecap = c->es.cap / 4 >= 2 ? c->es.cap / 4 - 2 : 0;
  for (const Buf* b : {&c->ei, &c->ej}) ecap = b->cap / 4 < ecap ? b->cap / 4 : ecap;
  c->est_allowed = true;  // (what the entry points that come here set for the call: see edge_build_ok)
  const bool will_build = edge_build_ok(c, p, n);
  c->est_allowed = false;
  if (!will_build) for (const Buf* b : {&c->ebi, &c->ebj}) ecap = b->cap / 4 < ecap ? b->cap / 4 : ecap;  // (the fused edge kernel writes no per-edge bases)
  const uint64_t kcap = c->wkey.cap / 4 < c->kcol.cap / 8 ? c->wkey.cap / 4 : c->kcol.cap / 8;
  // the last call's counts plus half, and the largest counts of the shape's recent calls plus a quarter (sc_ctx::E_hi)
  const bool young = c->hi_seen < sc_ctx::HI_YOUNG;
  uint64_t ecov = cover_of(c->E_last, c->E_hi, young), mcov = cover_of(c->M_last, c->M_hi, young);
  if (ecov > ecap) ecov = ecap;
  if (mcov > kcap) mcov = kcap;
  if (ecov < c->E_last || ecov < 4096 || ecov >= (1ull << 32) || mcov < c->M_last || mcov < p->max_triangles) return false;
  c->E_cov = ecov; c->M_cov = mcov;
  return true;
}

// the whole path the waiting way (what sc_register_device always did): complete on return.  The only caller of the phases
// that may prune by an ESTIMATED bound (it can repeat the call): a failed estimate comes back as SC_ESPEC from finalize_wait
// and has switched estimating off for this context, so the second pass certifies.
int register_waited(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p, float* d_Rt,
                    uint8_t* d_mask, sc_stats* stats) {
  c->spec_on = false;
  int rc = SC_OK;
  for (int pass = 0; pass < 2; pass++) {
    c->est_allowed = true;
    rc = hyp_begin(c, d_src, d_tgt, n, p, nullptr, 0, 1);
    if (!rc) rc = hyp_end(c, nullptr, c->key.as<uint64_t>(), stats);
    c->est_allowed = false;
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    rc = finalize_enqueue(c, c->key.as<uint64_t>(), 1, d_Rt, d_mask);
    if (!rc) rc = finalize_wait(c, stats);
    if (rc != SC_ESPEC) return rc;
  }
  c->last_error = "internal: a call was asked to repeat twice";
  return SC_EHIP;
}

}  // namespace

extern "C" {

// one more frame (a call of sc_register_device(_async) / sc_register, or sc_hypothesize_device + its finalize call) has come to its
// end with status rc: the context's cumulative counters (sc_debug_last)
static void count_frame(sc_ctx* c, int rc) {
  c->n_frames++;
  if (c->est_failed_call) c->n_est_fail++;
  else if (c->est_state == 1 && (rc == SC_OK || rc == SC_ENOHYP)) c->n_est_ok++;
  if (c->fast_state == 2 || (rc == SC_EBOUND && !c->est_failed_call)) c->n_fast_repeat++;  // host-free, void: repeated (here, or by the caller)
  else if (c->fast_state == 1) c->n_fast_ok++;
}

// the finalize call of a HOST-FREE sc_hypothesize_device: a failed validation is the caller's to repeat (every rank alike)
static int finalize_status(sc_ctx* c, int rc) {
  if (rc != SC_ESPEC) return rc;
  c->last_error += " — repeat sc_hypothesize_device and the finalize call without SC_FLAG_EST_BOUND";
  return SC_EBOUND;
}

int sc_finalize_gathered_device(sc_ctx* c, const uint64_t* d_keys, int n_pairs, float* d_Rt, uint8_t* d_mask,
                                sc_stats* stats) {
  if (!c || !d_keys || !d_Rt || !d_mask || n_pairs < 1 || n_pairs > 4096) return SC_EINVAL;
  if (c->pending) { c->last_error = "a call is outstanding on this context (sc_wait first)"; return SC_EINVAL; }
  if (!c->have_hyp) { c->last_error = "sc_finalize_device without a preceding sc_hypothesize_device"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = finalize_enqueue(c, d_keys, n_pairs, d_Rt, d_mask);
  if (rc) return rc;
  rc = finalize_status(c, finalize_wait(c, stats));
  count_frame(c, rc);
  return rc;
}

int sc_finalize_gathered_device_async(sc_ctx* c, const uint64_t* d_keys, int n_pairs, float* d_Rt, uint8_t* d_mask) {
  if (!c || !d_keys || !d_Rt || !d_mask || n_pairs < 1 || n_pairs > 4096) return SC_EINVAL;
  if (c->pending) { c->last_error = "a call is outstanding on this context (sc_wait first)"; return SC_EINVAL; }
  if (!c->have_hyp) { c->last_error = "sc_finalize_device without a preceding sc_hypothesize_device"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  memset(&c->pend_stats, 0, sizeof c->pend_stats);
  c->pend_stats.size = sizeof(sc_stats);
  const int rc = finalize_enqueue(c, d_keys, n_pairs, d_Rt, d_mask);
  if (rc) return rc;
  c->pending = true; c->pend_done = false; c->pend_finalize = true;
  return SC_OK;
}

int sc_register_device_async(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p,
                             float* d_Rt, uint8_t* d_mask) {
  if (!c || !d_src || !d_tgt || !d_Rt || !d_mask) return SC_EINVAL;
  if (c->pending) { c->last_error = "sc_register_device_async: a call is already outstanding on this context (sc_wait first)"; return SC_EINVAL; }
  int rc = check_params(p);
  if (rc) return rc;
  if (p->shard_world != 1) return SC_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  ENSURE(c, c->key, 64);
  c->pend_src = d_src; c->pend_tgt = d_tgt; c->pend_n = n; c->pend_p = *p; c->pend_Rt = d_Rt; c->pend_mask = d_mask;
  memset(&c->pend_stats, 0, sizeof c->pend_stats);
  c->pend_stats.size = sizeof(sc_stats);
  c->fast_state = 0;
  c->est_failed_call = false;
  if (fast_plan(c, n, p)) {
    // host-free: the whole chain is enqueued without looking at anything the GPU produces; sc_wait validates
    c->spec_on = true;
    c->est_allowed = true;  // (sc_wait repeats a call whose estimated pruning bound fails, like any other failed assumption)
    rc = hyp_begin(c, d_src, d_tgt, n, p, nullptr, 0, 1);
    if (!rc) rc = hyp_end(c, nullptr, c->key.as<uint64_t>(), &c->pend_stats);
    c->est_allowed = false;
    if (!rc) rc = finalize_enqueue(c, c->key.as<uint64_t>(), 1, d_Rt, d_mask);
    if (!rc) {
      c->pending = true; c->pend_done = false; c->pend_finalize = false;
      return SC_OK;
    }
    // The host-free enqueue failed (an allocation sized by the COVERS hit the workspace cap, a launch failed): the waited form of
    // the same call may well fit — "results are identical either way" (include/saccot.h) includes the status.  What was enqueued
    // so far runs out first: its kernels publish into the pinned words the waited call is about to arm.  
    c->spec_on = false; c->fast_ok = false;
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
  }
  rc = register_waited(c, d_src, d_tgt, n, p, d_Rt, d_mask, &c->pend_stats);
  if (rc != SC_OK && rc != SC_ENOHYP) return rc;
  c->pending = true; c->pend_done = true; c->pending_rc = rc; c->pend_finalize = false;
  return SC_OK;
}

int sc_wait(sc_ctx* c, sc_stats* stats) {
  if (!c) return SC_EINVAL;
  if (!c->pending) { c->last_error = "sc_wait without an outstanding sc_register_device_async / sc_finalize_gathered_device_async"; return SC_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  c->pending = false;
  if (c->pend_finalize) {  // sc_finalize_gathered_device's second half
    c->pend_finalize = false;
    const int frc = finalize_status(c, finalize_wait(c, &c->pend_stats));
    count_frame(c, frc);
    if (stats && stats->size == sizeof(sc_stats)) *stats = c->pend_stats;
    return frc;
  }
  int rc = c->pending_rc;
  if (!c->pend_done) {
    rc = finalize_wait(c, &c->pend_stats);
    if (rc == SC_ESPEC) {  // a count outgrew what the launches covered (or another fallback was needed): the waiting way
      rc = register_waited(c, c->pend_src, c->pend_tgt, c->pend_n, &c->pend_p, c->pend_Rt, c->pend_mask, &c->pend_stats);
      c->fast_state = 2;
    }
  }
  count_frame(c, rc);
  if (stats && stats->size == sizeof(sc_stats)) *stats = c->pend_stats;
  return rc;
}

int sc_register_device(sc_ctx* c, const float* d_src, const float* d_tgt, int64_t n, const sc_params* p,
                       float* d_Rt, uint8_t* d_mask, sc_stats* stats) {
  const int rc = sc_register_device_async(c, d_src, d_tgt, n, p, d_Rt, d_mask);
  return rc ? rc : sc_wait(c, stats);
}

int sc_register(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p, float R[9],
                float t[3], uint8_t* mask, sc_stats* stats) {
  if (!c || !src || !tgt || !R || !t || !mask || n < 3 || n > (1 << 24)) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  if (p->shard_world != 1) return SC_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  if (n <= (1 << 20)) {
    // Zero-copy form (up to 1 M correspondences = 24 MB in, 1 MB out over the host link): two memcpys on the host into a
    // pinned, device-mapped area; the staging kernel reads it, the finalize kernel writes (R, t) and the mask into the
    // other.  No hipMemcpy at all: a pageable-memory copy costs 10-20 us of driver time each (C2: 0.36 -> 0.31 ms).
    const size_t in_bytes = (size_t)n * 24, out_bytes = 64 + (size_t)n;
    auto grow = [&](void** p_, size_t* cap, size_t want) -> int {
      if (*cap >= want) return SC_OK;
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (*p_) { (void)hipHostFree(*p_); *p_ = nullptr; *cap = 0; }
      size_t sz = want + want / 4 + 4096;
      if (hipHostMalloc(p_, sz, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *p_ = nullptr; c->last_error = "hipHostMalloc failed"; return SC_ENOMEM; }
      *cap = sz;
      return SC_OK;
    };
    if ((rc = grow(&c->h_in, &c->h_in_cap, in_bytes))) return rc;
    if ((rc = grow(&c->h_out, &c->h_out_cap, out_bytes))) return rc;
    float* hs = static_cast<float*>(c->h_in);
    float* ht = hs + (size_t)n * 3;
    memcpy(hs, src, (size_t)n * 12);
    memcpy(ht, tgt, (size_t)n * 12);
    void *d_s = nullptr, *d_t = nullptr, *d_o = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(&d_s, hs, 0));
    HIPCHK(c, hipHostGetDevicePointer(&d_t, ht, 0));
    HIPCHK(c, hipHostGetDevicePointer(&d_o, c->h_out, 0));
    rc = sc_register_device(c, static_cast<float*>(d_s), static_cast<float*>(d_t), n, p, static_cast<float*>(d_o),
                            static_cast<uint8_t*>(d_o) + 64, stats);
    if (rc != SC_OK && rc != SC_ENOHYP) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (idle already on the private stream; a caller's stream may hold the refit)
    const float* Rt = static_cast<const float*>(c->h_out);
    memcpy(R, Rt, 36);
    memcpy(t, Rt + 9, 12);
    memcpy(mask, static_cast<const uint8_t*>(c->h_out) + 64, (size_t)n);
    return rc;
  }
  ENSURE(c, c->in_src, (size_t)n * 12);
  ENSURE(c, c->in_tgt, (size_t)n * 12);
  ENSURE(c, c->rt12, 64);
  ENSURE(c, c->mask, (size_t)n);
  HIPCHK(c, hipMemcpyAsync(c->in_src.p, src, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->in_tgt.p, tgt, (size_t)n * 12, hipMemcpyHostToDevice, c->stream));
  rc = sc_register_device(c, c->in_src.as<float>(), c->in_tgt.as<float>(), n, p, c->rt12.as<float>(),
                          c->mask.as<uint8_t>(), stats);
  if (rc != SC_OK && rc != SC_ENOHYP) return rc;
  float Rt[12];
  HIPCHK(c, hipMemcpyAsync(Rt, c->rt12.p, 48, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(mask, c->mask.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(R, Rt, 36);
  memcpy(t, Rt + 9, 12);
  return rc;
}

// ---- stage hooks ------------------------------------------------------------------------------------

int sc_compat_host(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p, float* S,
                   uint64_t* bits, uint32_t* deg) {
  if (!c) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false;
  c->timing = c->timing_hot = false; c->timing_one = -1;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  c->dv = derive(p);
  const bool dense = S != nullptr && !(p->flags & SC_FLAG_NO_DENSE_S);
  if (S && !dense) { c->last_error = "sc_compat_host: S requested together with SC_FLAG_NO_DENSE_S"; return SC_EINVAL; }
  if ((rc = host_to_planes(c, src, tgt, n, p))) return rc;
  if ((rc = run_compat(c, dense))) return rc;
  if ((rc = run_row_stats(c, false))) return rc;
  if ((rc = check_flag(c))) return rc;
  const size_t W = (size_t)c->ld >> 6;
  if (S)
    HIPCHK(c, hipMemcpy2DAsync(S, (size_t)n * 4, c->S.p, (size_t)c->ld * 4, (size_t)n * 4, (size_t)n,
                               hipMemcpyDeviceToHost, c->stream));
  if (bits) HIPCHK(c, hipMemcpyAsync(bits, c->bits.p, (size_t)n * W * 8, hipMemcpyDeviceToHost, c->stream));
  if (deg) HIPCHK(c, hipMemcpyAsync(deg, c->deg.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return SC_OK;
}

int sc_triangles_host(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p, uint32_t* tri,
                      uint32_t* key, uint32_t* t_eff, uint64_t* tri_total, uint64_t* edges) {
  if (!c || !tri || !t_eff) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false;
  c->timing = c->timing_hot = false; c->timing_one = -1;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  c->dv = derive(p);
  if ((rc = host_to_planes(c, src, tgt, n, p))) return rc;
  if ((rc = run_compat(c, false))) return rc;  // the ranked list needs the bit rows only
  if ((rc = run_row_stats(c, may_prune(p)))) return rc;
  sc_params pe = *p;
  pe.flags |= SC_FLAG_EXACT_TOTAL;  // the hook reports the 3-clique count of the whole graph
  if ((rc = run_triangles(c, &pe, true))) return rc;
  *t_eff = c->T_eff;
  if (tri_total) *tri_total = c->M_total;
  if (edges) *edges = c->E;
  if (c->T_eff) {
    // the hook returns the RANKED list; the hot path itself never sorts
    const size_t T = c->T_eff, sort_bytes = sort_temp_bytes(T);
    ENSURE(c, c->sortkey, T * 8);
    ENSURE(c, c->sorted, T * 8);
    ENSURE(c, c->sort_tmp, sort_bytes + 16);
    ENSURE(c, c->tri_rk, T * 12);
    ENSURE(c, c->key_rk, T * 4);
    launch_rank_order(c->tri.as<uint32_t>(), c->sel_key.as<uint32_t>(), c->T_eff, c->sortkey.as<uint64_t>(),
                      c->sorted.as<uint64_t>(), c->sort_tmp.p, sort_bytes, c->tri_rk.as<uint32_t>(),
                      c->key_rk.as<uint32_t>(), c->stream);
    HIPCHK(c, hipMemcpyAsync(tri, c->tri_rk.p, T * 12, hipMemcpyDeviceToHost, c->stream));
    if (key) HIPCHK(c, hipMemcpyAsync(key, c->key_rk.p, T * 4, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return SC_OK;
}

int sc_kabsch_host(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p,
                   const uint32_t* tri, uint32_t n_tri, float* Rt) {
  if (!c || !tri || !Rt) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  for (size_t k = 0; k < (size_t)n_tri * 3; k++) if ((int64_t)tri[k] >= n) return SC_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  if ((rc = host_to_planes(c, src, tgt, n, p))) return rc;
  if (n_tri == 0) return check_flag(c);
  ENSURE(c, c->tri, (size_t)n_tri * 12);
  ENSURE(c, c->rt_aos, (size_t)n_tri * 48);
  HIPCHK(c, hipMemcpyAsync(c->tri.p, tri, (size_t)n_tri * 12, hipMemcpyHostToDevice, c->stream));
  launch_kabsch_aos(points_of(c), c->tri.as<uint32_t>(), n_tri, c->rt_aos.as<float>(), c->stream);
  if ((rc = check_flag(c))) return rc;
  HIPCHK(c, hipMemcpyAsync(Rt, c->rt_aos.p, (size_t)n_tri * 48, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return SC_OK;
}

int sc_score_host(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p, const float* Rt,
                  uint32_t n_hyp, uint32_t* cnt, uint64_t* key) {
  if (!c || !Rt || !key) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  c->dv = derive(p);
  if ((rc = host_to_planes(c, src, tgt, n, p))) return rc;
  Shard sh;
  sh.T_eff = n_hyp; sh.block = 0x40000000u; sh.rank = 0; sh.world = 1; sh.n_local = n_hyp;
  sh.ld_local = (uint32_t)(((uint64_t)n_hyp + 255u) / 256u * 256u);
  ENSURE(c, c->key, 64);
  if (n_hyp) {
    ENSURE(c, c->rt_aos, (size_t)sh.ld_local * 48);  // (the scoring kernel may read whole 256-hypothesis groups)
    ENSURE(c, c->rt, (size_t)12 * sh.ld_local * 4);
    HIPCHK(c, hipMemcpyAsync(c->rt_aos.p, Rt, (size_t)n_hyp * 48, hipMemcpyHostToDevice, c->stream));
    if (sh.ld_local > n_hyp)  // the padding hypotheses read as zeros in both layouts
      HIPCHK(c, hipMemsetAsync(c->rt_aos.as<char>() + (size_t)n_hyp * 48, 0, (size_t)(sh.ld_local - n_hyp) * 48, c->stream));
    launch_rt_to_soa(c->rt_aos.as<float>(), n_hyp, sh.ld_local, c->rt.as<float>(), c->stream);
  }
  // the choice of the C2 kernel looks at the coordinate maxima the staging kernel publishes: wait for them (the path
  // proper never has to — it has polled later results of the same stream by the time it gets here)
  if (!c->tn.filter_blind && (rc = wait_word(c, 13))) return rc;
  c->sh = sh;
  if ((rc = decide_filter(c, p, sh))) return rc;
  uint32_t score_rows = 0;
  if ((rc = run_score(c, p, sh, &score_rows, false))) return rc;
  ENSURE(c, c->cnt, (size_t)(sh.ld_local ? sh.ld_local : 256) * 4);
  ENSURE(c, c->amx_pairs, argmax_scratch_bytes(sh.ld_local));
  launch_argmax(sh, c->partial.as<uint32_t>(), score_rows, nullptr, c->cnt.as<uint32_t>(), c->amx_pairs.as<uint64_t>(),
                &c->ctl.as<ControlBlock>()->amx_ticket, c->key.as<uint64_t>(), c->stream);  // positions in Rt ARE the rank indices here: single-stage key
  if ((rc = check_flag(c))) return rc;
  if (cnt && n_hyp) HIPCHK(c, hipMemcpyAsync(cnt, c->cnt.p, (size_t)n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(key, c->key.p, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return SC_OK;
}

int sc_mask_host(sc_ctx* c, const float* src, const float* tgt, int64_t n, const sc_params* p, const float Rt[12],
                 uint8_t* mask) {
  if (!c || !Rt || !mask) return SC_EINVAL;
  int rc = check_params(p);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  c->have_hyp = false;
  c->cap_bytes = p->max_workspace ? p->max_workspace : (64ull << 30);
  c->dv = derive(p);
  if ((rc = host_to_planes(c, src, tgt, n, p))) return rc;
  ENSURE(c, c->rt12, 64);
  ENSURE(c, c->mask, (size_t)n);
  HIPCHK(c, hipMemcpyAsync(c->rt12.p, Rt, 48, hipMemcpyHostToDevice, c->stream));
  launch_mask(points_of(c), c->rt12.as<float>(), c->dv.tau2, c->mask.as<uint8_t>(), c->stream);
  if ((rc = check_flag(c))) return rc;
  HIPCHK(c, hipMemcpyAsync(mask, c->mask.p, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  return SC_OK;
}

}  // extern "C"
