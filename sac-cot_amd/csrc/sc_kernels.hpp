// sc_kernels.hpp — launchers of the hand-written gfx950 kernels (one per SURVEY.md §8(a) row).
// Every launcher only enqueues on `st`; none allocates, frees or synchronises (capture-safe).
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace sc {

std::vector<uint32_t> compat_wg_map(int W);  // sc_compat.hip: which 64 x 64 block workgroup b of stage A takes (XCD-aware)

// Derived fp32 constants, computed once per call on the host in fp64.
struct Derived {
  float d_thr;         // sigma * sqrt(-2 ln t_cmp)
  float neg_inv2sig2;  // -1 / (2 sigma^2)
  float tau2;          // tau^2
  float min_len;
  float inv_tau2;      // 1 / tau^2   (score modes MSE / MAE, include/saccot.h)
  float inv_tau;       // 1 / tau
};

// Scheduling / fallback knobs of the kernels.  The shipped library reads NO environment variable: these defaults are
// what runs, and only sc_set_debug() (include/saccot.h, test and tuning hook) changes them, per context.  None of them
// can change a result — only the launch geometry or which of two bit-identical code paths runs.
struct Tuning {
  bool no_events = false;          // row-walking count / key kernels instead of the event list
  uint64_t event_cap = 0;          // forced event capacity (0: the context's own, grown after an overflow)
  size_t compact_self_max = 4096;  // tiles up to which compact_write adds up the tile counts by itself
  size_t scan_self_max = 4096;     // tiles up to which the scan's down-sweep adds up the block sums by itself
  uint32_t cnt_blocks = 0, keys_blocks = 0, sel_blocks = 0;  // grid sizes (0: automatic)
  int tg_count = 8, tg_keys = 8, tg_sample = 0;              // lanes per edge (tg_sample 0: by row width)
  int tg_events = 0;               // lanes per edge of the event-recording counting pass (0: by row width)
  uint32_t sample_mode = 0;        // pruning sample: 0 by size (launch_sample_hist), 1 every stride-th edge, 2 the heaviest edges (both CERTIFY their bound)
  bool no_edge_build = false;      // row statistics and edge list as separate launches (not launch_edge_build)
  bool compat_linear_order = false;  // stage A's 64 x 64 blocks in index order (r03) instead of the XCD-aware order
  bool no_estimate = false;        // never prune by an ESTIMATED bound (sc_tri.hip 3c): always one of the certifying samples
  uint32_t est_margin_pct = 0;     // the estimate aims at the (pct / 100 x T)-th key (0: 200; tests force failures with a small one)
  uint32_t sample_blocks = 0;      // grid of the heaviest-edge sample (0: one block per 256 edges)
  bool rows_unfused = false;       // row_stats and the scan(s) of the row counts as separate launches (round 1's form)
  uint64_t sample_edges = 0;       // edges of the pruning sample (0: automatic, ~5T/8)
  uint32_t score_split = 0;        // share (of 256) of the hypotheses scored on the matrix pipe
  uint32_t score_filter = 0;       // C2, inlier count: 0 = by size and scale (plain kernel / linear filter / Gram filter), 1 = plain kernel, 2 = linear filter, 3 = Gram filter
  uint32_t filter_splits = 0;      // grid.y of the filter (0: by size)
  uint32_t filter_queue_cap = 0;   // entries of the filter's global queue (0: by size; tests force overflows with a small one)
  uint32_t filter_lds_queue = 0;   // entries of a wave's LDS queue, 64 .. 256 (0: 256)
  uint32_t filter_variant = 0;     // body of the filter kernel (sc_score.hip): 0 default, bit-identical scheduling variants, >= 16 timing-only ablations
  uint32_t gram_kappa_q4 = 0;      // the Gram filter's cut: hypotheses whose reach stays under (value / 16) tau' count as NEAR the reference (0 = GX_KAPPA = 8; 1 = practically no cut)
  bool gram_ref_late = false;      // the Gram filter's reference frame is voted after the selection, in a launch of its own (what every path but the hot one does anyway), instead of under the counting pass
  bool no_fast = false;            // sc_register_device never enqueues host-free (always waits for stage B's two counts)
  bool gram_guard_fail = false;    // the matrix-pipe probe reports a violation (tests of the guard)
  bool filter_blind = false;       // the host decides C2's kernel WITHOUT the coordinate maxima (as if they had not arrived yet)
  bool compat_one_phase = false;   // exact chain on every pair of an interior tile
  int compat_rows = 0;             // tile height of stage A: 0 = by size (16 below 10 000 correspondences, 32 from there), 16, 32, 64
  uint32_t compat_store_mode = 0;  // 0: by size; bit 0: force 4-byte S stores, bit 2: force 16-byte, bit 1: non-temporal
};

// LbArgs: what a look-back launch needs from its caller.  desc: the tile descriptors (8 bytes per tile and value) in a
// persistent area that ONLY look-back launches write — zeroed once, and again whenever the 12-bit epoch wraps; epochs
// 1 .. 4095 are handed out one per launch, so words of earlier launches read as "nothing yet".  ticket / err: two
// words that are NEVER descriptor storage (a ticket that aliased an old descriptor would hand out garbage tile
// indices): the ticket is zero between launches (the last tile of a launch resets it).
struct LbArgs {
  uint32_t* ticket = nullptr;
  uint32_t* err = nullptr;
  uint64_t* desc = nullptr;
  uint32_t epoch = 0;
};

// Device view of the padded SoA point planes: px py pz qx qy qz, each `ld` floats (ld = roundup(n,64)),
// zero-filled beyond n — followed, at planes + 6 * ld, by an AoS copy of 8 floats per correspondence
// (px py pz qx qy qz 0 0; 32-byte aligned) for the consumers that fetch one whole correspondence at a time.
struct Points {
  const float* planes;  // (6 + 8) * ld
  int n;
  int ld;
};

// ---- input staging -----------------------------------------------------------------------------
// user layout (AoS n x 3 or SoA 3 x n) -> padded planes; sets *bad_flag != 0 when a value is not finite.
// bad_flag may be host-pinned memory: it is only touched (atomicOr) when a non-finite value is found.
// zero / zero_words: a buffer (the per-call control block) the kernel clears on the way — saves a memset launch.
// coord_max (optional, 2 x u32): atomicMax of the bit patterns of max |src coordinate| and max |tgt coordinate|; with it:
// coord_max_next (the pair the NEXT call uses: cleared by the last block), mx_ticket (a zeroed u32, left zero; nullptr: the rows
// of coord_part stay unreduced — launch_compat's stat_part) and
// host_max (pinned u64: receives max|tgt| << 32 | max|src| once every block is done).
// The coordinate statistics are FX_MX_WORDS u32 (coord_max, WRITTEN — not accumulated — by the last block of the launch):
// [0] max |src coordinate|, [1] max |tgt coordinate| (bit patterns), [2 + c] / [8 + c]: float_key of the largest / of minus
// the smallest value of coordinate c (px py pz qx qy qz): the bounding boxes.  coord_part: 16 u32 per block of the launch
// (stage_part_words(ld)), scratch.  host_box (pinned, 6 x u64, optional): (key of -min) << 32 | key of max per
// coordinate, written before host_max.
constexpr int FX_MX_WORDS = 16;
inline size_t stage_part_words(int ld) { return (size_t)((ld + 255) / 256) * 16; }
void launch_stage_points(const float* d_src, const float* d_tgt, int n, int ld, int layout, float* planes,
                         uint32_t* bad_flag, uint32_t* zero, uint32_t zero_words, uint32_t* coord_max, uint32_t* coord_part,
                         uint32_t* mx_ticket, uint64_t* host_max, uint64_t* host_box, hipStream_t st,
                         uint32_t* zero2 = nullptr, uint32_t zero2_words = 0);  // zero2: a second buffer cleared on the way
// order-preserving map float -> u32 (all finite floats and infinities; 0 is below every float) and back
__host__ __device__ inline uint32_t float_key(float f) {
  union { float f; uint32_t u; } x; x.f = f;
  return (x.u & 0x80000000u) ? ~x.u : (x.u | 0x80000000u);
}
__host__ __device__ inline float float_unkey(uint32_t k) {
  union { float f; uint32_t u; } x; x.u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return x.f;
}
// can C2's filter bound its error at this tau?  (the host-side twin of the test every wave of the filter makes; maxima as
// the staging kernel published them, ~0 = not known: assume yes)
bool filter_in_range(uint64_t host_max, float tau2);

// ---- stage A: compat_graph -----------------------------------------------------------------------
// S: n x ld fp32 (row-major, symmetric, zero diagonal / pad columns); bits: n x (ld/64) u64;
// rows [row0, row1) (row0 a multiple of 64): the whole matrix by symmetric tiles for [0, n), one-sided tiles for a
// row block (then S, if given, holds the rows of the block only).  S == nullptr: adjacency bits only.
// degp (optional, whole-matrix form only; n u32, ZEROED): also accumulates deg+[i] = edges (i, j) with j > i (atomics).
// wg_map / wg_map_len (optional, whole-matrix form): the XCD-aware order of the 64 x 64 blocks, compat_wg_map(ld / 64) on the
// device — the launch then has wg_map_len workgroups.
// stat_part / stat_out (optional): the staging launch was given no ticket (mx_ticket == nullptr) and left its per-workgroup rows
// of coordinate statistics unreduced: one wave of this launch reduces them into the FX_MX_WORDS words.
void launch_compat(const Points& pts, const Derived& dv, float* S, uint64_t* bits, int row0, int row1, const Tuning& tn,
                   hipStream_t st, uint32_t* degp = nullptr, const uint32_t* wg_map = nullptr, uint32_t wg_map_len = 0,
                   const uint32_t* stat_part = nullptr, uint32_t* stat_out = nullptr);
// deg[i] = edges of i; degp[i] = edges (i,j) with j > i; wpre: n x (ld/64) u32, set bits of row i in words [0,w).
// zero_rows (optional): an n x W u64 matrix cleared on the way (the pruned bit matrix of stage B).
// rowcost (optional, n u32): per-row estimate of stage B's work (see row_stats_kernel), for launch_shard_split.
void launch_row_stats(const Points& pts, const uint64_t* bits, uint32_t* deg, uint32_t* degp, uint32_t* wpre,
                      uint64_t* zero_rows, uint32_t* rowcost, hipStream_t st);
// row_stats and the prefixes over the rows in one launch (decoupled look-back over 64-row tiles): also edge_off (n + 1
// CSR row offsets), ebase (n per-row CSR bases), cost_pre (optional, n + 1: prefix of the row costs) and the edge count
// into host_total.  state: row_stats_scan_state_bytes(n) of the caller's look-back state area; epoch: this launch's.
size_t row_stats_scan_state_bytes(int n);
void launch_row_stats_scan(const Points& pts, const uint64_t* bits, uint32_t* deg, uint32_t* degp, uint32_t* wpre,
                           uint64_t* zero_rows, uint64_t* edge_off, uint32_t* ebase, uint64_t* cost_pre, const LbArgs& lb,
                           uint64_t* host_total, hipStream_t st);
// this rank's contiguous, equally heavy row range (own_row[0..1]) and its CSR edge range (own_edge[0..1])
// from the exclusive prefix of rowcost (n + 1 entries) — device-side, identical on every rank.
void launch_shard_split(const uint64_t* cost_pre, const uint64_t* edge_off, int n, uint32_t rank, uint32_t world,
                        uint32_t* own_row, uint64_t* own_edge, hipStream_t st);

// ---- exclusive scan u32 -> u64 (out has n+1 entries; out[n] = total) -----------------------------
// host_total (optional): host-pinned u64 that also receives the total, written by the kernel itself — the host
// reads it after its next stream synchronise, with no copy kernel in between.
// Extras a scan can do on its way (all optional):
//   range (device, [lo, hi)): outside it every input is known to be zero — tiles wholly outside are neither read nor
//     written (sharded stage B: the triangle counts of the edges other ranks enumerate); out[n] is still the total;
//   deg / degp / ebase: also ebase[i] = (u32) out[i] - (deg[i] - degp[i]), the CSR base of row i (launch_edge_fill).
//   lb.epoch != 0: the single-pass (decoupled look-back) form.
struct ScanExtra {
  LbArgs lb;
  const uint64_t* range = nullptr;
  const uint32_t* deg = nullptr;
  const uint32_t* degp = nullptr;
  uint32_t* ebase = nullptr;
};
size_t scan_temp_bytes(size_t n);
void launch_scan_u32(const uint32_t* in, size_t n, uint64_t* out, void* temp, const Tuning& tn, hipStream_t st,
                     uint64_t* host_total = nullptr, const ScanExtra* x = nullptr);
// two arrays of the same length in one go (one launch when n is small); in1/out1 may be null; host_total and x0: of
// array 0
void launch_scan_u32_pair(const uint32_t* in0, uint64_t* out0, const uint32_t* in1, uint64_t* out1, size_t n,
                          void* temp, const Tuning& tn, hipStream_t st, uint64_t* host_total = nullptr,
                          const ScanExtra* x0 = nullptr, const ScanExtra* x1 = nullptr);

// ---- stage B: triangles_topT ---------------------------------------------------------------------
struct Graph {
  const uint64_t* bits;  // n x W
  const float* S;        // n x ld
  const uint32_t* deg;
  const uint32_t* degp;  // edges to higher indices
  const uint32_t* wpre;  // n x W word-prefix popcounts of `bits`
  int n, ld, W;
};
// CSR edge list of the upper triangle, rows ascending, columns ascending: ei/ej/es (es = S[i][j]).
// ebase[i] (u32, modular): CSR index of edge (i,k), k > i, is ebase[i] + wpre[i][k/64] + popc(bits[i][k/64] below k).
// ebi[e] / ebj[e]: the bases of both ends of edge e (so an edge is fetched in one memory level).
// ebase_ready: the per-row CSR bases were written by the scan of deg+ (ScanExtra, tiled form) and are READ here;
// otherwise they are derived on the fly and this kernel writes them.  scan_writes_ebase(n) tells which.
bool scan_writes_ebase(size_t n);
void launch_edge_fill(const Graph& g, const Points& pts, const Derived& dv, const uint64_t* edge_off, uint32_t* ei,
                      uint32_t* ej, float* es, uint32_t* ebase, bool ebase_ready, uint32_t* ebi, uint32_t* ebj,
                      uint64_t cap, uint32_t* es_hist, hipStream_t st);  // es_hist: see launch_sample_hist (optional)
// The hot path's form (r04; n <= 20 480, weight ranking, estimated pruning bound): row statistics (wpre), CSR offsets and
// bases from g.degp — which launch_compat accumulated —, the strong-bit rows cleared and the edge list (ei / ej / es; NO ebi /
// ebj: launch_tri_count_events looks the bases up), in one launch.  host_total (pinned) receives the edge count, which also
// lands in edge_off[n].
bool edge_build_fits(int n);
void launch_edge_build(const Graph& g, const Points& pts, const Derived& dv, uint64_t* zero_rows, uint64_t* edge_off, uint32_t* ebase,
                       uint32_t* ei, uint32_t* ej, float* es, uint64_t cap, uint64_t* host_total, hipStream_t st,
                       uint64_t* live_range = nullptr);  // live_range (optional, device, 2 x u64): receives [0, edge count)
// tcnt[e] = #k > j adjacent (in `mbits`) to both ends of edge e = (i,j); edges with es[e] < *smin count 0
// (smin == nullptr: no pruning, mbits = g.bits).
void launch_tri_count(const Graph& g, const uint64_t* mbits, const float* es, const float* smin, const uint32_t* ei,
                      const uint32_t* ej, uint64_t E, uint32_t* tcnt, const uint64_t* own, const Tuning& tn,
                      hipStream_t st);  // own (optional, device): [lo, hi) of the edges this rank enumerates
// Compact list of the strong edges (those of the pruned graph), written by the pruning kernel in ST_SHARDS regions of
// `cap` entries (fill[r] = entries of region r; ST_SHARDS zeroed counters of the control block).  list == nullptr: off.
constexpr int ST_SHARDS = 256;
struct StrongList {
  uint32_t* list;
  uint32_t* fill;
  uint32_t cap;
  // 0: region r takes the pruning kernel's blocks with block % 256 == r.  > 0: region r takes `region_blocks` CONSECUTIVE
  // blocks (256 edges each), so a region is a contiguous range of edge ids — a rank of the sharded stage B then walks only the
  // regions its own edge range touches (the pruning kernel also zeroes every tcnt entry in this form)
  uint32_t region_blocks;
};
uint32_t strong_list_cap(uint64_t E);
size_t strong_list_bytes(uint64_t E);

// Certified pruning (weight ranking), two launches:
//  launch_sample_hist: the triangles of every R-th edge go into `hist` (256 u32, zeroed by the caller); `part` of
//    `parts`: this launch takes every parts-th SAMPLED edge starting at `part` (one process per GPU: the histograms of
//    all parts are summed before launch_prune_bits sees them).  key_floor: a value at or below the smallest possible
//    triangle weight.
//  launch_prune_bits: derives the strong-edge threshold *smin (device float; -1 = nothing certified) and *klb (key of
//    the bound or 0) from the histogram, builds the strong upper-triangle bit matrix `mbits` (n x W, already zero), and
//    with sl.list set also compacts the strong edges and zeroes tcnt of the weak ones.
// E_dev (optional, device): the true edge count when the host launches before it knows it (host-free path, sc_capi.hip):
// the kernels then work on min(E, *E_dev) edges — E is what the grids and arrays cover; E_hint (0: E): the edge count the
// host-side choices (sample form, stride) are made with.
void launch_sample_hist(const Graph& g, const uint32_t* ebi, const uint32_t* ebj, const uint32_t* ei,
                        const uint32_t* ej, const float* es, uint64_t E, uint64_t want, float key_floor, uint32_t part,
                        uint32_t parts, uint32_t* hist, uint32_t* es_hist, const Tuning& tn, hipStream_t st,
                        const uint64_t* E_dev = nullptr, uint64_t E_hint = 0);
// The ESTIMATING sample (r04, sc_tri.hip 3c): one sampled 64-column word in `rate` of every edge's row pair, its triangles'
// keys into `hist` — a uniform 1-in-rate sample of ALL the graph's triangles.  launch_prune_bits then takes the bin where
// rate x (count from the top) reaches margin x T as the bound: an estimate, not a certificate — the select verifies it
// (SelectState::want_req) and the caller repeats the call with a certifying sample when it was too high.
struct SamplePlan {
  bool estimate;       // true: launch_sample_estimate / hist_want below; false: launch_sample_hist (certifying), hist_want = T
  uint32_t rate;       // power of two
  uint64_t hist_want;  // what launch_prune_bits looks for in the histogram
};
SamplePlan sample_plan(uint64_t want, bool allow_estimate, const Tuning& tn, uint64_t E = 0, int W = 0);  // E, W: the graph (0: not known: rate <= 64)
void launch_sample_estimate(const Graph& g, const uint32_t* ebi, const uint32_t* ebj, const uint32_t* ei, const uint32_t* ej,
                            const float* es, uint64_t E, float key_floor, uint32_t rate, uint32_t* hist, const Tuning& tn,
                            hipStream_t st, const uint64_t* E_dev = nullptr, const uint32_t* ebase = nullptr,  // ebase: with ebi == ebj == nullptr
                            uint4* cand = nullptr, unsigned long long* cand_slot = nullptr);  // cand (optional, sample_estimate_blocks() entries): the best-keyed triangle each workgroup sampled {key bits, i, j, k} — the voters of stage C2's reference frame (sc_gramref.hpp)
uint32_t sample_estimate_blocks(uint64_t E, const Tuning& tn);  // workgroups launch_sample_estimate uses
uint32_t sample_candidate_blocks(uint64_t E, const Tuning& tn);  // ... of which the first so many leave a candidate behind
// es_hist: PR_HCOPIES x 256 words (control block): the weight histogram of the heaviest-edge sample, filled by launch_edge_fill
// launch_sample_hist ALWAYS accumulates into PR_HCOPIES x 256 words (`hist` = the control block's copies);
// launch_hist_reduce sums them into one 256-bin histogram (the exchanged form); launch_prune_bits reads either
// (hist_is_copies).
void launch_hist_reduce(const uint32_t* copies, uint32_t* out, hipStream_t st);
void launch_prune_bits(const Graph& g, const uint32_t* hist, bool hist_is_copies, const uint32_t* ei, const uint32_t* ej, const float* es,
                       uint64_t E, uint64_t want, float key_floor, uint64_t* mbits, float* smin, uint32_t* klb,
                       const StrongList& sl, uint32_t* tcnt, const uint64_t* own, hipStream_t st,
                       const uint64_t* E_dev = nullptr,   // E_dev: as above; with sl.list the tcnt entries of [*E_dev, E) are zeroed too
                       bool logbins = false,              // hist comes from launch_sample_estimate (logarithmic bins of 3.0 - key)
                       bool trimmed = false);             // (with E_dev) the scan that follows skips the counts beyond *E_dev: the workgroups there leave at once
// sharded stage B, after the certificate: work estimate per row of the PRUNED graph (strong_rowcost_kernel), and — in one
// single-block launch — its prefix and this rank's row / edge range (the rule of launch_shard_split)
void launch_strong_rowcost(const Graph& g, const uint64_t* mbits, uint32_t* rowcost, hipStream_t st);
void launch_cost_split(const uint32_t* rowcost, const uint64_t* edge_off, int n, uint32_t rank, uint32_t world,
                       uint32_t* own_row, uint64_t* own_edge, hipStream_t st);

// Event list of stage B (sc_tri.hip 2b): one record per non-zero member word of a strong edge, SoA, split into
// EV_SHARDS regions of shard_cap records; fill[shard] = records appended to that region.
struct EventList {
  uint64_t* m;       // member word (bits = common neighbours k of the edge inside this 64-column word)
  uint32_t* wi;      // word offset of row i: i * W + w   (into bits / wpre)
  uint32_t* wj;      // word offset of row j
  uint32_t* a;       // weight ranking: ebase[i];  degree ranking: deg[i] + deg[j]
  uint32_t* b;       // weight ranking: ebase[j]
  uint32_t* e;       // edge index
  uint32_t* rb;      // rank of the word's first triangle inside its edge
  uint32_t* fill;    // EV_SHARDS counters
  uint64_t shard_cap;
  uint32_t* overflow;  // host-pinned flag, set when a region is full
  int W;
};
struct GramRefJob;
size_t event_bytes(uint64_t capacity);
EventList event_list(void* buf, uint64_t capacity, int W, uint32_t* fill, uint32_t* overflow_host);
// counting pass that also emits the events (replaces launch_tri_count when an event buffer is available)
void launch_tri_count_events(const Graph& g, const uint64_t* mbits, const StrongList& sl, const uint32_t* ebi,
                             const uint32_t* ebj, const uint32_t* ei, const uint32_t* ej, uint64_t E, int rank_mode,
                             uint32_t* tcnt, const EventList& ev, const Tuning& tn, hipStream_t st,
                             const uint64_t* own = nullptr,   // own (optional, device): [lo, hi) of the edges this rank enumerates
                             const uint32_t* ebase = nullptr,  // with ebi == ebj == nullptr: the per-row CSR bases (launch_edge_build)
                             const GramRefJob* ref = nullptr); // one extra workgroup votes for stage C2's reference frame (sc_gramref.hpp)

// Radix-select state.  Lives in the context's control block, which the staging kernel zeroes per call.
// r05: no workgroup of a select round stays behind to "pick": a round only adds its keys into hist[r], and the NEXT launch —
// round r + 1, or the compaction's counting kernel after the last round — resolves that histogram in every workgroup's prologue
// (select_resolve: 16 loads per thread + one block scan, ~1 us, overlapped with its first key loads); its workgroup 0 stores the
// resolved state for the launches after it.  The release + ticket + last workgroup's pass this replaces were ~4 us of a round's 8.
struct SelSnap {           // the select after some rounds: keys in [lo, lo + 2^wbits) still undecided, `above` keys above it
  uint32_t lo, wbits;      // valid once started != 0 (before: the window is the key range [kmin, kmax])
  uint32_t started, done;  // done: the window is one key value, lo = k*
  uint64_t want;           // number of keys to keep (clamped to what the window holds)
  uint64_t above;          // keys strictly above the window
  uint64_t need_eq;        // done: how many keys == k* to keep (lowest ordinals first)
  uint64_t pad;
};
struct SelectState {
  uint32_t kmin, kmax;     // key range
  uint32_t kstar, done;    // the result (compact_count_kernel's workgroup 0 writes it; every later kernel reads these)
  uint64_t want;           //   keys kept
  uint64_t need_eq;        //   how many keys == kstar among them
  uint64_t want_req;       // != 0: the window's floor is a bound that must have this many keys at or above it (whoever resolves a
                           // round reports a shortfall: an ESTIMATED pruning bound that was too high, sc_tri.hip 3c)
  uint64_t pad;
  SelSnap st[4];           // st[r]: the state BEFORE round r (st[0]: key_range_kernel / the key kernel's preset / merge_prepare_kernel)
  uint32_t hist[3][4096];  // hist[r]: round r's bins; all zero between selects (compact_write_kernel clears them behind the select)
};
static_assert(offsetof(SelectState, hist) % 16 == 0, "SelectState::hist must be 16-byte aligned");

constexpr int PR_HCOPIES = 4;  // global copies of the pruning-sample histogram (block b adds into copy b % 4): with one
                               // copy a thousand blocks' adds into the same 256 words serialise (C2 sample 34 -> 27 us);
                               // with 16 the readers' 16 loads per bin cost prune_bits what the sample gained
// Per-call control block (device memory, zeroed by the staging kernel at the start of every call).
struct ControlBlock {
  uint32_t ev_fill[1024];    // event-list region fill counters (EV_SHARDS)
  uint32_t prune_hist[PR_HCOPIES * 256];  // sampled key histogram of the certified pruning, PR_HCOPIES partial copies
  uint32_t es_hist[PR_HCOPIES * 256];     // edge-weight histogram (which edges the sample takes), same layout
  uint32_t st_fill[256];     // strong-edge list region fill counters (ST_SHARDS)
  float smin;                // strong-edge threshold (written by prune_bits_kernel)
  uint32_t amx_ticket;       // blocks-finished counter of score_argmax_kernel
  uint32_t klb;              // key of the certified lower bound of the pruning (0: none)
  uint32_t pad_fin[2];
  uint32_t own_row[2];       // sharded stage B: this rank's row range [lo, hi) ...
  uint32_t pad0[9];
  uint64_t key2[2];          // internal winner key pair (sc_register_device)
  unsigned long long fin_word;  // finalize_kernel: workgroups finished << 32 | rank count so far — ONE returning atomic per workgroup (left zero)
  uint64_t own_edge[2];      // ... and its CSR edge range (launch_shard_split)
  uint64_t live_edges[2];    // host-free calls: [0, edges of the graph) — written by launch_edge_build; the scan of the per-edge counts skips the tiles beyond it
  uint64_t pad1[1];
  SelectState sel;
  // stage C2's reference frame (sc_gramref.hpp): slot v = the best (key bits << 32 | workgroup) among the workgroups = v (mod 64) of the
  // estimating sample — its voter v is that workgroup's candidate triangle
  alignas(8) unsigned long long ref_slot[64];
};
static_assert(offsetof(ControlBlock, sel) % 16 == 0, "ControlBlock::sel must be 16-byte aligned");

// key of every triangle, in ordinal (lexicographic i,j,k) order: wkey[toff[e] + r].
// blk_minmax: 2 * 8192 u32 scratch (per-block key min / max, reduced into s->kmin / s->kmax; s->want = want).
void launch_tri_keys(const Graph& g, const uint64_t* mbits, const float* smin, const uint32_t* ebase,
                     const uint32_t* ei, const uint32_t* ej, const float* es, const uint64_t* toff, uint64_t E,
                     int rank_mode, uint32_t* wkey, uint2* kcol, uint32_t* blk_minmax, SelectState* s,
                     uint64_t want, const uint64_t* own, const Tuning& tn, hipStream_t st);  // kcol[ordinal] = {the triangle's third vertex, its edge id}
// keys from the event list (replaces launch_tri_keys when no region overflowed)
void launch_tri_keys_events(const Graph& g, const float* es, const uint64_t* toff, int rank_mode,
                            const EventList& ev, uint32_t* wkey, uint2* kcol, uint32_t* blk_minmax,
                            SelectState* s, uint64_t want, const uint32_t* klb, uint64_t E, uint64_t cap,
                            const Tuning& tn, hipStream_t st, bool check_bound = false);  // cap: entries of wkey / kcol (writes beyond are dropped); want is clipped to toff[E]
// check_bound: with a pruning bound in *klb the select must find `want` keys at or above it (SelectState::want_req)
// klb != nullptr (weight ranking, every edge weight >= 2/3): the kernel presets the select window to
// [*klb or 2.0, 3.0] and no key-range pass runs; two select rounds then always suffice.
// A key array as the select / compaction kernels see it.  Plain (seg_len == 0): M keys at base.  Segmented (the
// all-gathered candidate blobs of sharded stage B): M = segments x seg_len logical entries, seg_len a multiple of 1024;
// segment s holds valid[s * valid_stride] real keys at base + s * seg_stride, the rest of it reads as 0.
struct KeyView {
  const uint32_t* base;
  uint64_t M;
  uint64_t seg_len;
  uint64_t seg_stride;    // u32 words between segment starts
  const uint64_t* valid;  // real keys of segment s: valid[s * valid_stride]
  uint64_t valid_stride;  // u64 words
  // plain view only, optional (device): the true key count when the host launched before it knew it — the kernels then
  // look at min(M, *M_dev) keys; M is what the grids cover (host-free path, sc_capi.hip)
  const uint64_t* M_dev;
};
KeyView plain_view(const uint32_t* wkey, uint64_t M);
// `rounds` launches (one histogram of <= 12 key bits each; round r + 1 — and launch_compact_count after the last — resolves round r:
// see SelectState) find the exact threshold key
// host_short (optional, pinned u64): set to 1 when SelectState::want_req keys were promised above the window's floor and
// fewer are there
void launch_select_rounds(const KeyView& view, SelectState* s, int rounds, const Tuning& tn, hipStream_t st,
                          uint64_t* host_short = nullptr);
// compaction of the selected keys in ordinal order
size_t compact_blocks(uint64_t M);
// rounds: what launch_select_rounds was given (the counting kernel resolves the last round's histogram; host_short as there)
void launch_compact_count(const KeyView& view, SelectState* s, int rounds, uint32_t* blk_gt, uint32_t* blk_eq,
                          hipStream_t st, uint64_t* host_short = nullptr);
void launch_compact_write(const KeyView& view, SelectState* s, const uint32_t* blk_gt,
                          const uint32_t* blk_eq, const uint64_t* off_gt, const uint64_t* off_eq,
                          uint64_t* sel_ord, uint32_t* sel_key, uint64_t n_sel, hipStream_t st);  // n_sel: entries sel_ord / sel_key hold

// ---- sharded stage B: the candidate blob a rank sends, and the merge of the gathered blobs ----
constexpr int CAND_HDR_WORDS = 32;  // u64 words: [0] triangles enumerated, [1] entries sent, [2] local threshold key,
                                    // [3] / [4] a key range containing every key sent, [5] 1 = the rank had more
                                    // candidates than the blob holds (its list is cut at its local threshold key)
struct CandBlob {
  uint64_t* hdr;
  uint32_t* keys;  // cap entries, (i,j,k) ascending
  uint4* recs;     // cap entries {i, j, k, key}
  size_t cap;
};
// Entries per blob.  A rank can contribute up to T triangles to the global top-T, but with equally heavy row ranges it
// contributes about T / world: the blob holds twice that (at least 4096), times 2^level — `level` (sc_params.
// shard_cand_level) is raised by the caller after SC_ERETRY, i.e. when the merge found that a cut list could have
// mattered (launch_merge_check); negative levels shrink the blob (tests).  Never more than T rounded up to 1024.
size_t cand_cap(uint32_t T, uint32_t world, int level);
size_t cand_blob_bytes(size_t cap);  // bytes per rank
CandBlob cand_blob(void* blob, size_t cap);
// this rank's selection (sel_ord / sel_key, n_max an upper bound of its length) -> blob
void launch_cand_emit(const uint64_t* sel_ord, const uint32_t* sel_key, const uint2* kcol, const uint32_t* ei,
                      const uint32_t* ej, const SelectState* sel, const uint64_t* toff, uint64_t E, uint32_t n_max,
                      uint64_t want_requested, const CandBlob& b, hipStream_t st);
// sums the gathered headers, arms `sel` for the merge select; host_out[0] <- T_eff (polled), host_out[1] <- triangles
void launch_merge_prepare(const void* blobs, size_t blob_bytes, uint32_t world, uint32_t T, bool fast, const uint32_t* klb,
                          SelectState* sel, uint64_t* host_out, hipStream_t st,
                          uint64_t* host_short = nullptr);  // host_short (pinned): set when *klb is an ESTIMATED bound and fewer than T candidates came
KeyView cand_view(const void* blobs, size_t blob_bytes, uint32_t world, size_t cap);
// After the merge select: a rank whose list was cut sent everything above its local threshold key k_r; the merged
// selection is exact iff the merged threshold k* lies strictly above k_r for every such rank (everything it did not send
// is then below k*).  Otherwise *host_flag = 1 (pinned; 0 is written otherwise): the call must be repeated with bigger blobs.
void launch_merge_check(const void* blobs, size_t blob_bytes, uint32_t world, const SelectState* sel, uint64_t* host_flag,
                        hipStream_t st);
// ranked order: sortkey ascending = (key desc, ordinal asc).  Implemented with rocPRIM (sc_sort.hip).
size_t sort_temp_bytes(size_t n);
void launch_sort_u64(const uint64_t* in, uint64_t* out, size_t n, void* temp, size_t temp_bytes, hipStream_t st);
// selected position -> (i,j,k); the list stays in ordinal ((i,j,k) ascending) order
void launch_tri_decode(const uint32_t* ei, const uint32_t* ej, const uint2* kcol, const uint64_t* sel_ord, uint32_t T,
                       uint32_t* tri, hipStream_t st);
// ranked order (key desc, ordinal asc) of the ordinal-ordered list: only the stage hook needs it
void launch_rank_order(const uint32_t* tri, const uint32_t* sel_key, uint32_t T, uint64_t* sortkey, uint64_t* sorted,
                       void* sort_tmp, size_t sort_bytes, uint32_t* tri_ranked, uint32_t* key_ranked, hipStream_t st);

// ---- stage C ---------------------------------------------------------------------------------------
struct Shard {
  uint32_t T_eff;   // ranked triangles in total
  uint32_t block;   // dealing granularity
  uint32_t rank, world;
  uint32_t n_local; // hypotheses of this rank
  uint32_t ld_local;// roundup(n_local, 256): plane stride of RtSoA / partial counts
};
uint32_t shard_local_count(uint32_t T_eff, uint32_t block, uint32_t rank, uint32_t world);
// Where stage C finds triangle g of the selected list: the selection itself (position -> ordinal -> {third vertex,
// edge} -> edge ends) — the hot path never materialises the T x 3 list (only the stage hook does, launch_tri_decode).
struct TriSource {
  const uint64_t* sel_ord;
  const uint2* kcol;
  const uint32_t* ei;
  const uint32_t* ej;
  // sharded stage B: sel_ord[g] is a logical position in the gathered candidate blobs; its record {i, j, k, key}
  // is cand_recs[(pos / cand_seg) * cand_stride + pos % cand_seg]   (cand_recs == nullptr: the local form above)
  const uint4* cand_recs;
  uint64_t cand_seg, cand_stride;  // entries per blob; uint4 units between blob record arrays
  // host-free calls only (0: unchecked): what a looked-up {vertex, edge} must stay below — a call that turns out to have
  // been launched on wrong assumptions (and is repeated) may find anything in the selection, and must not follow it
  uint32_t lim_vertex, lim_edge;
  uint64_t lim_ord;
};
// C1: RtSoA[c * ld_local + l], c = 0..11, for the local hypotheses of the shard (global rank index derived).
// RtAoS (optional): also 12 consecutive floats per local hypothesis (ld_local x 12), for the lane = correspondence kernel
// tile_job (optional): the filter's tile kernel rides in the same launch as extra workgroups (C1 alone leaves most of the
// chip idle: T / 256 workgroups), which saves a launch on the critical path.
struct FilterTileJob;
// t_eff_dev (optional, device u64): the true length of the selection when the host launched with sh.T_eff = the requested
// T before it knew (host-free path): positions at or beyond it are treated as padding and never looked up
void launch_kabsch(const Points& pts, const TriSource& ts, const Shard& sh, float* RtSoA, float* RtAoS,
                   const FilterTileJob* tile_job, hipStream_t st, const uint64_t* t_eff_dev = nullptr);
// C1 on an explicit triangle list to AoS T x 12 (stage hook)
void launch_kabsch_aos(const Points& pts, const uint32_t* tri, uint32_t T, float* Rt, hipStream_t st);
// AoS T x 12 -> SoA planes (stage hook for sc_score_host)
void launch_rt_to_soa(const float* Rt, uint32_t T, uint32_t ld_local, float* RtSoA, hipStream_t st);
// C2: inlier counts.  partial: n_chunks * ld_local u32 scratch.
// The plain C2 kernel: lane = hypothesis with the points in LDS (every score mode, the matrix-pipe share).
// score_chunks: point chunks that launch will use (rows of `partial`).
uint32_t score_chunks(int n, uint32_t ld_local);
// score_mode: 0 inlier count, 1 truncated squared residual, 2 truncated absolute residual (include/saccot.h)
// ev0 / ev1 (both or neither): start and stop timestamps taken from the dispatch packets of the stage's first and last kernel
// (hipExtLaunchKernelGGL) — no extra packet on the stream, unlike a bracket of hipEventRecord calls (~4.5 us each on this part)
void launch_score(const Points& pts, const float* RtSoA, const float* RtAoS, const Shard& sh, const Derived& dv,
                  int score_mode, uint32_t* partial, const Tuning& tn, hipStream_t st, hipEvent_t ev0 = nullptr,
                  hipEvent_t ev1 = nullptr);
// C2, inlier count, by filter + exact fix-up (sc_score.hip): which calls use it, what it needs, and its two launches.
// The tile (fp16 image of the correspondences) depends on the points only: launch_filter_tile runs once per call, any
// time after launch_stage_points (whose atomicMax fills mx_cur); mx_cur / mx_next are two u32 pairs that alternate
// from call to call (the tile kernel clears the next call's).  partial: fp.splits rows of ld_local counts.
// Two filters share the machinery (tile of the correspondences, queue of undecided tests, recount bitmap, exact pass):
//   mode 1, "linear":  the MFMA gives the three residual components of 8 hypotheses x 32 correspondences, the vector
//                      pipe squares and tests them (4.75 vector instructions per test);
//   mode 2, "Gram":    the MFMA gives the SQUARED residual itself — |R p + t - q|^2 expanded into a 48-term dot product
//                      of per-correspondence features and per-hypothesis coefficients, 32 hypotheses x 32 correspondences
//                      per three chained MFMAs — and the vector pipe only tests its sign (1.6 instructions per test).
//                      Squaring by inner products cancels: usable while tau is not too small against the clouds' extent.
struct FilterPlan {
  uint32_t mode;  // 1 linear, 2 Gram
  uint32_t windows, splits, n_waves, rows, queue_cap;
  size_t tile_bytes, state_bytes, coef_bytes;  // coef_bytes: the Gram filter's per-hypothesis coefficients and its two permutations (0 for the linear one)
};
// Gram filter: what its waves load instead of computing it (made once per call: launch_kabsch's threads, or launch_gram_coef).
// Rows are PERMUTED: hypotheses near the call's reference frame first (sc_score.hip, "the cut"); so are the tile's rows.
struct GramFrame;
struct GramCoef {
  void* A;            // 64 bytes per row: fp16 halves of its 16 coefficients
  float* C;           // accumulator start value per row
  float* W;           // shell width per row (0: not filtered)
  uint32_t* flag;     // bit 0 filtered, bit 1 recount, bits 8.. log2 of the largest alpha
  uint32_t* hperm;    // coefficient row -> hypothesis of this rank's shard
  uint32_t* pperm;    // tile row -> correspondence
  GramFrame* frame;   // the call's frame and the four class counters (in the filter's state buffer)
};
GramCoef gram_coef_view(void* buf, uint32_t ld_local, void* frame);
// which kernel stage C2 runs: 0 = plain fp32 kernel, 1 = linear filter, 2 = Gram filter.  By score mode, size, the knobs of
// sc_debug and — unless forced — by whether tau is on a scale the filter can bound (host_max / host_box: the coordinate
// maxima and bounding boxes the staging kernel published; ~0 = not known: assume the linear filter applies, never Gram).
int score_filter_mode(int score_mode, const Tuning& tn, int n, uint32_t ld_local, uint64_t host_max, const uint64_t* host_box,
                      float tau2);
FilterPlan filter_plan(int n, uint32_t ld_local, const Tuning& tn, uint32_t mode);
struct FilterTileJob {  // what the tile kernel needs (filter_tile_job fills it)
  uint32_t rows; const uint32_t* mx_cur; uint32_t* mx_next; void* tile; void* info; uint32_t* zero; uint32_t zero_words; uint32_t mode;
  GramCoef coef; float tau2;  // mode 2: the Kabsch threads of the same launch also write their hypotheses' coefficients
};
FilterTileJob filter_tile_job(const FilterPlan& fp, const uint32_t* mx_cur, uint32_t* mx_next, void* tile, void* state,
                              void* coef, uint32_t ld_local, float tau2, void* frame);
// Gram filter: the call's reference frame (sc_gramref.hpp; `frame`: gram_frame_bytes() of device memory, per context).  One
// workgroup votes for it among 64 hypotheses and clears the class counters — BEFORE the tile and the coefficients are made.  In a
// launch of its own (launch_gram_ref: RtSoA = the hypotheses if they exist already — stage hook —, else null: they are solved
// from the selection `ts`), or, on the hot path, as an extra workgroup of the counting pass (launch_tri_count_events, `ref`) fed
// by the candidate triangles the estimating sample leaves behind (launch_sample_estimate, `cand`).
size_t gram_frame_bytes();
struct GramRefJob;
GramRefJob gram_ref_job(const Points& pts, const uint32_t* mx, float tau2, const Tuning& tn, void* frame);  // (source left empty)
void launch_gram_ref(const Points& pts, const TriSource& ts, const Shard& sh, const float* RtSoA, const uint64_t* t_eff_dev,
                     const uint32_t* mx, float tau2, const Tuning& tn, void* frame, hipStream_t st);
// the coefficients on their own (stage hook sc_score_host, which has no Kabsch launch)
void launch_gram_coef(const float* RtSoA, const Shard& sh, float tau2, const GramCoef& coef, hipStream_t st);
void launch_filter_tile(const Points& pts, const FilterTileJob& job, hipStream_t st);  // on its own (stage hook sc_score_host)
// RtAoS: 12 consecutive floats per hypothesis (what the exact pass loads; launch_kabsch / the stage hook write them)
void launch_score_filter(const Points& pts, const float* RtSoA, const float* RtAoS, const Shard& sh, const Derived& dv,
                         const FilterPlan& fp, const void* tile, void* state, void* coef, void* frame, uint32_t* partial, const Tuning& tn,
                         hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                         hipEvent_t ev_mid = nullptr);  // ev_mid (optional): stop timestamp of the FILTER kernel's own dispatch (ev0 .. ev_mid = that kernel alone)
// Run-time probe of the matrix pipe's accumulation arithmetic (the model the Gram filter's bound assumes; sc_score.hip):
// blocking, ~1e6 cancelling dot products.  scratch: 16 bytes of device memory.  worst_units: largest |hardware - exact| /
// largest term seen, in units of 2^-24 (the bound assumes 18.5; gram_guard_limit() is what a context tolerates).
hipError_t gram_guard_probe(void* scratch, hipStream_t st, float* worst_units, bool* subnormals_kept, uint32_t* compared);
float gram_guard_limit();
bool filter_ablations_built();  // -DSC_ABLATIONS: the scheduling / timing-only variants of the filter kernels exist (Tuning::filter_variant)
// diagnostics (sc_debug_last): what the filter of the last launch handed to the exact pass.  Blocking copies on `st`.
hipError_t filter_read_counters(const void* state, const FilterPlan& fp, hipStream_t st, uint64_t* undecided, uint64_t* recounts);
// Gram filter: {near correspondences, near hypotheses, hypotheses in the grid, the reference's position in the shard (~0: none), its vote x 256}
hipError_t filter_read_frame(const void* frame, hipStream_t st, uint32_t out[5]);
// Winner key pair key2[0..1] (written, not accumulated: no zeroing needed):
//   key2[0] = max over hypotheses with count > 0 of  (count << 32) | second,   second = sel_key[g] (the triangle's
//             ranking key) or, when sel_key == nullptr, 0xFFFFFFFF - g;
//   key2[1] = max of (0xFFFFFFFF - g) over the hypotheses attaining key2[0]  (only when sel_key != nullptr).
// With g the position in the ordinal-ordered list this picks: most inliers, then best ranking key, then lowest
// (i,j,k) — exactly "ties -> best-ranked triangle" of SURVEY §8a, without ever sorting the list.
// cnt (n_local u32): the per-hypothesis counts (the stage hook returns them).  pairs: argmax_scratch_bytes() of
// per-block (key, position) pairs; ticket: a zeroed u32 in the control block (left zero).
size_t argmax_scratch_bytes(uint32_t ld_local);
uint32_t argmax_blocks(uint32_t ld_local);  // workgroups of launch_argmax = pairs it leaves when key2 == nullptr
void launch_argmax(const Shard& sh, const uint32_t* partial, uint32_t n_chunks, const uint32_t* sel_key,
                   uint32_t* cnt, uint64_t* pairs, uint32_t* ticket, uint64_t* key2, hipStream_t st);
// C3: winner decode + re-solve + mask.  Rt12 receives R (row-major) and t; identity / zero mask when key2[0] == 0.
// sel_key / T: the ordinal-ordered ranking keys (for the winner's rank index); host_out (pinned, 2 x u64) receives
// [1] = rank index << 32 | position (all ones: the key pair decodes to nothing of the selection) and then, released, [0] = key2[0].
// key2: npairs key pairs (all-gathered, one per rank; 1 = already reduced); key_out (2 x u64, optional) receives the
// reduced pair.
// sh / RtSoA: this rank's shard and the (R,t) planes phase 1 produced — a locally scored winner is looked up there.
// dp (optional; host-free calls): what the call's earlier kernels would have published to the host one by one — stage B's two counts
// (device words; they go to dp->host[0] as ONE word: edges | triangles << 32, ~0 if either needs more than 32 bits) and, with_stats,
// the staging kernel's coordinate statistics (FX_MX_WORDS words -> [13] maxima, [16 .. 21] boxes) — written by THIS kernel, before
// the winner.  Those kernels then publish nothing: a system-scope store costs the kernel that makes it ~0.5 us.
struct DeferredPub {
  const uint64_t* dev_edges;
  const uint64_t* dev_triangles;
  const uint32_t* coord_max;
  unsigned long long* host;
  int with_stats;
};
void launch_finalize(const Points& pts, const TriSource& ts, const Shard& sh, const float* RtSoA,
                     const uint32_t* sel_key, uint32_t T, const uint64_t* key2, int npairs, uint64_t* key_out, float tau2, float* Rt12, uint8_t* mask,
                     unsigned long long* fin_word, uint64_t* host_out, hipStream_t st, const DeferredPub* dp = nullptr);
// SURVEY §8f-2 (SC_FLAG_REFINE): fp64 least-squares refit of Rt12 over the inlier mask; no-op when key2[0] == 0 or
// fewer than 3 inliers.  scratch: refine_scratch_bytes(n).
size_t refine_scratch_bytes(int n);
void launch_refine(const Points& pts, const uint8_t* mask, const uint64_t* key2, double* scratch, float* Rt12,
                   hipStream_t st);
// mask of an explicit hypothesis (stage hook)
void launch_mask(const Points& pts, const float* Rt12, float tau2, uint8_t* mask, hipStream_t st);

}  // namespace sc
