/*
 * saccot.h — C ABI of libsaccot.so: the SAC-COT compatibility-triangle sample-consensus hot path
 * on AMD Instinct MI355X (gfx950).
 *
 * What this replaces in the reference
 * -----------------------------------
 * The upstream tree (ytuhzq/SAC-COT) is a single file, /root/reference/README.md:1-2, which names the
 * algorithm ("SAC-COT: Sample Consensus by Sampling Compatibility Triangles in Graphs for 3-D Point Cloud
 * Registration") and ships no code, no FFI and no tests.  There is therefore no reference interface to
 * cite beyond README.md:2; the boundary below is the one fixed by BASELINE.json `north_star`
 * ("correspondence-in / (R,t,inlier-mask)-out", "thin C-ABI layer") and SURVEY.md §8(b).
 * Every entry point says which SURVEY §8(a) row it implements.
 *
 * Conventions
 * -----------
 *  - plain C types only; no C++ exception crosses this boundary; every function returns an int status
 *    (0 = SC_OK, negative = error) unless stated otherwise;
 *  - "host" entry points take host pointers and do the H2D/D2H copies themselves; "_device" entry points
 *    take device pointers (hipMalloc / torch CUDA tensors) and enqueue on the context's stream;
 *  - a context (`sc_ctx`) is bound to ONE GPU and ONE stream (one process per GPU).  Calls on one context
 *    must be serialised by the caller; distinct contexts are independent; there is no global mutable state;
 *  - device workspace is owned by the context, grows on demand, is never shrunk, and is freed by
 *    sc_destroy();  the library never keeps a caller pointer past the return of the call;
 *  - there is NO CPU fallback in this library: without a usable HIP device sc_create() fails with SC_EHIP.
 */
#ifndef SACCOT_H
#define SACCOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_VERSION_MAJOR 0
#define SC_VERSION_MINOR 8   /* 0.8: sc_debug_info grew cumulative counters of how a context's frames ran (saccot_debug.h); every entry refuses a context with an outstanding call; a host-free enqueue that does not fit the workspace cap runs the waited way.  0.7: sc_finalize_gathered_device_async (+ sc_wait), sc_hypothesize_device with SC_FLAG_EST_BOUND host-free on a repeated shape.  0.6: SC_FLAG_EST_BOUND also on sc_hypothesize_device (SC_EBOUND from the finalize call); SC_FLAG_SHARD_AB (sc_register_multi replicates stages A and B on small graphs unless told otherwise); sc_debug / sc_debug_info grew (the Gram filter's frame and cut: saccot_debug.h).  0.5: sc_register_device_async / sc_wait (host-free enqueue), SC_FLAG_EST_BOUND / SC_EBOUND (sharded stage B pruned by an estimated bound), sc_stats.bytes_moved, the debug hooks moved to
                                saccot_debug.h; 0.4: sc_debug_last / sc_debug_info, sc_debug.filter_blind; 0.3: sc_set_debug (no environment variables), SC_FLAG_NO_DENSE_S, sc_shard_* (stages A and B sharded); 0.2: SC_FLAG_TIMING_HOT,
                                SC_STREAM_DEFAULT, sc_hypothesize_begin/end_device, sc_finalize_gathered_device */

/* status codes */
#define SC_OK        0
#define SC_EINVAL   -1   /* bad argument: n < 3, null pointer, non-finite input, bad params.size ...        */
#define SC_ENOMEM   -2   /* device or host allocation failed / workspace cap exceeded                       */
#define SC_EHIP     -3   /* HIP runtime error (sc_last_error() has the string)                              */
#define SC_ERCCL    -4   /* RCCL error or librccl.so.1 not loadable (sc_create_multi / sc_register_multi)   */
#define SC_ENOHYP   -5   /* no compatibility triangle / every inlier count is 0: R = I, t = 0, mask = 0     */
#define SC_ETOOMANY -6   /* the graph has more triangles than the workspace cap can rank (see max_workspace),  */
                         /* or 2^32 or more edges (edge ids are 32-bit)                                      */

#define SC_EBOUND   -8   /* calls made with SC_FLAG_EST_BOUND only: the ESTIMATED pruning bound was too high (the merged           */
                        /* candidates hold fewer than T keys above it) — or, sc_hypothesize_device, a host-free enqueue did not */
                        /* cover this input's counts: nothing was returned; repeat the call(s) WITHOUT the flag                */
#define SC_ERETRY   -7   /* sharded stages A + B only: a rank's candidate blob was too small for this input (its    */
                         /* list was cut at a key the merged threshold does not clear).  Outputs are not valid;    */
                         /* repeat the call on every rank with sc_params.shard_cand_level raised by one (every     */
                         /* rank sees the same blobs, so every rank returns it together).  sc_register_multi does  */
                         /* this by itself.                                                                        */

/* Limits: 3 <= n <= 2^24; max_triangles <= 2^32 - 256; the compatibility graph must have fewer than 2^32 edges
 * (SC_ETOOMANY otherwise; with the default 64 GiB workspace cap a dense graph runs out of workspace long before). */

/* point layouts for `src` / `tgt` */
#define SC_AOS 0   /* N x 3 row-major: x0 y0 z0 x1 y1 z1 ...                                   */
#define SC_SOA 1   /* 3 planes of N: x[0..N) y[0..N) z[0..N)  (a MATLAB N x 3 column-major array) */

/* triangle ranking */
#define SC_RANK_WEIGHT 0   /* w = (s_ij + s_ik) + s_jk, fp32, descending                     */
#define SC_RANK_DEGREE 1   /* deg_i + deg_j + deg_k, u32, descending                         */

/* hypothesis scoring (SURVEY §8f-2 `score_mode`).  The winner is the hypothesis with the largest score, ties as before
 * (best ranking key, then lowest (i,j,k)); the inlier MASK is always the test |R p + t - q| < tau.  The truncated
 * scores are sums of per-correspondence integers (10 fractional bits), so they are exact, order-free sums:
 *   SC_SCORE_MSE:  sum over n of  floor(1024 * max(0, 1 - d2_n / tau^2))      (MSAC: truncated squared residual)
 *   SC_SCORE_MAE:  sum over n of  floor(1024 * max(0, 1 - d_n / tau)),  d_n = sqrt(d2_n)  (truncated absolute residual)
 * with d2_n the canonical squared residual, 1 / tau^2 and 1 / tau rounded once from fp64, the product by fma. */
#define SC_SCORE_COUNT 0   /* number of inliers (default)                                      */
#define SC_SCORE_MSE   1
#define SC_SCORE_MAE   2

/* flags */
#define SC_FLAG_TIMING       1u /* record a HIP event pair around every stage and fill sc_stats.us_* (each record  */
                                /* costs ~5 us of stream time: diagnostics, not for the timed loop)               */
#define SC_FLAG_TIMING_HOT  16u /* only the dominant kernel: us_score (2 records per call)                          */
#define SC_FLAG_TIMING_ONE  64u /* ONE stage bracket (2 records per call), chosen by SC_TIMING_STAGE(k) in the flags: the   */
                                /* call runs its hot path (speculative launches on), so per-stage times taken one stage per   */
                                /* pass are times of the code that is actually timed end to end                                */
#define SC_TIMING_STAGE(k)  (((uint32_t)(k) & 15u) << 8) /* k: 0 staging, 1 compat, 2 triangles, 3 kabsch, 4 score, 5 argmax, 6 mask */
#define SC_FLAG_EXACT_TOTAL  2u /* sc_stats.tri_total = 3-cliques of the WHOLE graph (one extra counting pass);  */
                                /* default: 3-cliques of the pruned graph the top-T search actually enumerated   */
#define SC_FLAG_REFINE       8u /* after C3, replace (R,t) by the fp64 least-squares refit over the winner's inlier  */
                                /* mask; the mask itself stays the fp32 winner's                   */
#define SC_FLAG_NO_PRUNE     4u /* disable the certified pruning of stage B (results are identical either way)    */
#define SC_FLAG_EST_BOUND  128u /* phase API sc_shard_* only (sc_register / sc_register_device / sc_register_multi do this by themselves): */
                                /* stage B prunes by a bound ESTIMATED from a 1-in-64 sample of the graph's triangles instead of a      */
                                /* certified one.  EVERY rank takes the whole (cheap) sample, so sc_shard_edges_device leaves the same    */
                                /* histogram on every rank and the all-reduce after it MUST BE SKIPPED — three collectives per call, not  */
                                /* four.  The merge verifies the bound; when it was too high the finalize call returns SC_EBOUND on every  */
                                /* rank (nothing returned): repeat the call without this flag.  Results are identical either way.         */
                                /* Also taken by sc_hypothesize_device (stages A and B replicated on every rank; NOT by the _begin / _end   */
                                /* pair, whose shared histogram is a certifying sample's): sc_finalize_device / _gathered_device then      */
                                /* returns SC_EBOUND when the select found the bound too high — on every rank alike (the estimate is a      */
                                /* function of the input).  A HOST-FREE enqueue that outgrew its covers comes back as SC_EBOUND too, and     */
                                /* whether a rank enqueues host-free, and what its launches cover, follows from the history of ITS context:  */
                                /* ranks whose contexts have seen the same calls in the same order (bench.py's, sc_register_multi's) fail    */
                                /* together; a caller whose ranks may differ (a context recreated or warmed up differently) must AGREE on   */
                                /* the status before it repeats — any rank's SC_EBOUND means every rank repeats (sc_register_multi does).    */
#define SC_FLAG_SHARD_AB 4096u /* sc_register_multi only: shard stages A and B over the devices at EVERY size.  By default it does so from   */
                                /* 8192 correspondences on; below, every device runs stages A and B for the whole job (pruned by the      */
                                /* estimated bound) and scores its share — one 16-byte exchange per call instead of three collectives.    */
                                /* Results are identical either way.                                                                    */
#define SC_FLAG_NO_DENSE_S  32u /* stage A writes only the adjacency bit rows, not the dense n x n weight matrix S:   */
                                /* nothing after stage A reads S (edge weights are recomputed from the points), so    */
                                /* every result is identical; sc_compat_host returns S only without this flag         */

typedef struct sc_ctx sc_ctx;

/* All tunables of the path (SURVEY §8a `sc_params`).  POD; `size` must be sizeof(sc_params). */
typedef struct sc_params {
  uint32_t size;            /* = sizeof(sc_params); versioning                                          */
  float    sigma;           /* rigidity scale: s_ij = exp(-d^2 / (2 sigma^2)), d = | |pi-pj| - |qi-qj| | */
  float    t_cmp;           /* edge iff s_ij >= t_cmp  <=>  d <= sigma*sqrt(-2 ln t_cmp); in (0,1)       */
  float    tau;             /* inlier distance: |R p + t - q| < tau                                     */
  float    min_len;         /* edge additionally needs |pi-pj| >= min_len and |qi-qj| >= min_len        */
  uint32_t max_triangles;   /* T: hypotheses scored = top-T ranked compatibility triangles              */
  int32_t  rank_mode;       /* SC_RANK_WEIGHT / SC_RANK_DEGREE                                          */
  int32_t  layout;          /* SC_AOS / SC_SOA                                                          */
  int32_t  shard_rank;      /* this GPU's rank in [0, shard_world)                                      */
  int32_t  shard_world;     /* number of GPUs sharing the T hypotheses (1 = no sharding)                */
  uint32_t shard_block;     /* ranked triangles are dealt round-robin in blocks of this many (0 -> 1024) */
  uint32_t flags;           /* SC_FLAG_*                                                                */
  uint64_t max_workspace;   /* cap in bytes on the device workspace (0 -> 64 GiB)                       */
  int32_t  score_mode;      /* SC_SCORE_COUNT / SC_SCORE_MSE / SC_SCORE_MAE                             */
  int32_t  shard_cand_level;/* sharded A + B: a candidate blob holds max(2T/world, 4096) << level entries (<= T):    */
                            /* 0, raised by one after SC_ERETRY; negative values shrink the blob (tests)             */
} sc_params;

/* Per-call statistics (all optional: pass NULL).  Times are device times from HIP events on the
 * context's stream (each bracket minus the cost of one event record, calibrated once per context and stream), only filled when SC_FLAG_TIMING / SC_FLAG_TIMING_HOT is set, and delivered by the call
 * that ends the path (sc_register, sc_register_device, sc_finalize_device): sc_hypothesize_device never waits
 * for the GPU at its end, so the us_* fields of ITS stats stay 0. */
typedef struct sc_stats {
  uint32_t size;            /* = sizeof(sc_stats)                                                       */
  uint32_t n;               /* correspondences                                                          */
  uint64_t edges;           /* undirected edges of the compatibility graph                              */
  uint64_t tri_total;       /* 3-cliques enumerated (whole graph with SC_FLAG_EXACT_TOTAL / NO_PRUNE)   */
  uint32_t tri_kept;        /* T_eff = min(T, tri_total)                                                */
  uint32_t tri_scored;      /* hypotheses scored by THIS rank                                           */
  uint32_t best_rank;       /* rank index (0-based) of the winning triangle in the ranked list          */
  uint32_t best_count;      /* its score: the inlier count, or the truncated score of params.score_mode   */
  float    us_stage;        /* input staging (layout -> padded planes, finiteness check)                */
  float    us_compat;       /* stage A: the compat_rows kernel alone                                    */
  float    us_triangles;    /* stage B: every kernel of it plus its two 8-byte read-backs               */
  float    us_trikeys;      /* stage B: the tri_keys kernel alone (part of us_triangles)                */
  float    us_kabsch;       /* stage C1                                                                 */
  float    us_score;        /* stage C2: the score kernel alone                                         */
  float    us_argmax;       /* stage C2: partial-count reduction + arg-max key                          */
  float    us_mask;         /* stage C3: winner re-solve + mask                                         */
  float    us_total;        /* sum of the above                                                         */
  uint64_t workspace_bytes; /* device bytes currently held by the context                               */
  uint64_t bytes_moved;     /* ALGORITHMIC bytes this call's kernels read and wrote in device memory, from the counts of   */
                            /* the call (SURVEY §5 / §8d formulas: stage A 4n^2 + n^2/8 + 24n (n^2/8 + 24n without S),   */
                            /* stage B n^2/8 + 20 E + 12 M + 16 T_eff, stage C 52 T_scored + 24 n + n): a yardstick for    */
                            /* achieved-bandwidth figures, not a hardware counter                                          */
} sc_stats;

/* ---- library ---------------------------------------------------------------------------------- */
int         sc_version(void);                 /* (major << 16) | minor                               */
const char* sc_strerror(int status);          /* static string, never NULL                           */
void        sc_default_params(sc_params* p);  /* size set, sigma = tau = min_len = 0.1, t_cmp = 0.9,
                                                 T = 50000, weight ranking, AOS, no sharding          */

/* ---- context ---------------------------------------------------------------------------------- */
int         sc_create(int device, sc_ctx** out);      /* binds `device`, creates a private stream    */
void        sc_destroy(sc_ctx* ctx);                   /* frees the workspace; NULL is a no-op        */
int         sc_set_stream(sc_ctx* ctx, void* hip_stream); /* enqueue on a caller stream (e.g. torch's
                                                 current stream) instead of the private one, so that the
                                                 caller's own work on that stream (an all-reduce of d_key,
                                                 a copy of d_mask) is ordered with the kernels.  NULL
                                                 restores the private stream, which is NON-blocking: nothing
                                                 orders it against other streams.  The device's default
                                                 (null) stream has no handle of its own: pass
                                                 SC_STREAM_DEFAULT for it (torch reports it as 0)        */
#define SC_STREAM_DEFAULT ((void*)1)
const char* sc_last_error(const sc_ctx* ctx);          /* last HIP error text seen by this context    */

/* Test / tuning hooks (sc_set_debug, sc_debug_last) are NOT part of the drop-in surface: include/saccot_debug.h. */

/* ---- the drop-in entry point: correspondences in, (R, t, inlier mask) out ------------------------
 * north_star: "keeping the reference's correspondence-in / (R,t,inlier-mask)-out function signature".
 * Runs A (compat graph) -> B (ranked triangles) -> C1 (Kabsch) -> C2 (score, arg-max) -> C3 (mask) on the
 * GPU.  src/tgt: n points each, fp32, `params->layout`; row m of src corresponds to row m of tgt.
 * R: 3x3 row-major, t: 3, mask: n bytes (0/1), all caller-allocated HOST memory.  q ~ R p + t.
 * Requires shard_world == 1 (multi-GPU callers use the two-phase form below). */
int sc_register(sc_ctx* ctx, const float* src, const float* tgt, int64_t n, const sc_params* params,
                float R[9], float t[3], uint8_t* mask, sc_stats* stats);

/* Same, with every buffer already resident in HBM (d_Rt: 12 floats = R row-major then t).  Output visibility as
 * for sc_finalize_device below: complete on return with the private stream, stream-ordered with a caller stream. */
int sc_register_device(sc_ctx* ctx, const float* d_src, const float* d_tgt, int64_t n,
                       const sc_params* params, float* d_Rt, uint8_t* d_mask, sc_stats* stats);

/* The same call in two halves, for callers that register a STREAM of frames: sc_register_device_async enqueues the
 * whole path on the context's stream and returns without waiting for the GPU; sc_wait delivers the status and the
 * statistics of that call (and is where the host first looks at anything the GPU produced).  Between the two the host
 * is free — e.g. to enqueue the next frame on a SECOND context bound to the same stream (sc_set_stream): the GPU then
 * runs the frames back to back and never waits for the host.  At most ONE call may be outstanding per context
 * (SC_EINVAL otherwise); d_src / d_tgt must stay valid and unchanged until sc_wait returns, d_Rt / d_mask are complete
 * when it does (as for sc_register_device).
 * How it can return early: a call whose shape (n, parameters) equals the previous call's on this context is enqueued
 * "host-free" — its launches are sized by what the previous call needed (plus slack) and read the two data-dependent
 * counts of stage B (edges, triangles of the pruned graph) from device memory instead of from the host.  sc_wait
 * validates: had a count outgrown what the launches covered (or had any other of the waiting path's fallbacks been
 * needed), it repeats the call the waiting way before it returns — results are identical either way, bit for bit.
 * The first call on a context, and every call whose shape differs from the one before, simply waits inside
 * sc_register_device_async as sc_register_device always did.  sc_register_device is async + wait. */
int sc_register_device_async(sc_ctx* ctx, const float* d_src, const float* d_tgt, int64_t n,
                             const sc_params* params, float* d_Rt, uint8_t* d_mask);
int sc_wait(sc_ctx* ctx, sc_stats* stats);

/* ---- two-phase form for one-process-per-GPU sharding --------------------------------
 * Phase 1: A and B replicated, C1+C2 on this rank's blocks of the top-T list; writes this rank's winner key
 * PAIR to d_key (device, 2 x u64 = 16 bytes):
 *     d_key[0] = (inlier_count << 32) | ranking_key_of_the_triangle      (0 = no hypothesis with an inlier)
 *     d_key[1] = 0xFFFFFFFF - position of that triangle in the replicated top-T list, taking the LOWEST
 *                position among this rank's hypotheses that attain d_key[0]
 * The caller reduces over ranks in two steps (both 8-byte MAX all-reduces; RCCL through torch.distributed in
 * this repo's host layer, see sac-cot_amd/shard.py — any transport works):
 *     K0 = max_r d_key_r[0];   every rank whose own d_key[0] != K0 sets its d_key[1] = 0;   K1 = max_r d_key_r[1]
 * and stores (K0, K1) back into d_key.  The winner is thus: most inliers, then best ranking key, then lowest
 * (i,j,k) — "ties -> best-ranked triangle" of SURVEY §8a, decided without sorting the T hypotheses.
 * Phase 2: every rank decodes the same winner from the reduced pair, re-solves its (R,t) from its own
 * replicated list and builds the mask.  Returns SC_ENOHYP when d_key[0] is 0.
 * Synchronisation: sc_hypothesize_device returns with its last kernels still queued (d_key is valid in stream order:
 * enqueue the reduction on the same stream, or synchronise it).  sc_finalize_device returns once the winner is known
 * to the host (status, stats); on the context's private stream it also waits for d_Rt / d_mask, on a stream given
 * with sc_set_stream they are complete in THAT stream's order, like any other work the caller enqueues there. */
int sc_hypothesize_device(sc_ctx* ctx, const float* d_src, const float* d_tgt, int64_t n,
                          const sc_params* params, uint64_t* d_key, sc_stats* stats);
int sc_finalize_device(sc_ctx* ctx, const uint64_t* d_key, float* d_Rt, uint8_t* d_mask, sc_stats* stats);

/* Phase 2 on the all-GATHERED key pairs: d_keys holds n_pairs pairs (rank r's pair at d_keys[2r], d_keys[2r+1]), e.g. the
 * output of ONE all-gather of the 16-byte pairs; the reduction described above (a lexicographic max) runs inside the
 * finalize kernel.  One collective per call instead of two dependent ones; sc_finalize_device is the n_pairs = 1 case. */
int sc_finalize_gathered_device(sc_ctx* ctx, const uint64_t* d_keys, int n_pairs, float* d_Rt, uint8_t* d_mask,
                                sc_stats* stats);
/* The same in two halves (0.7), for ranks that register a STREAM of frames: the finalize kernel is enqueued and the call returns;
 * sc_wait(ctx, stats) delivers what sc_finalize_gathered_device returns.  Between the two the rank's host thread is free — e.g. to
 * run sc_hypothesize_device, the exchange and this call for the job's NEXT frame on a second context bound to the same stream, so
 * that the GPU never waits for a winner's way to the host.  With SC_FLAG_EST_BOUND a sc_hypothesize_device call that repeats the
 * last call's shape on its context is itself enqueued without a host wait (as sc_register_device_async's calls are); its
 * validation happens in the finalize call / sc_wait, and a failed one is one more reason for SC_EBOUND — on every rank alike.
 * (The statistics sc_hypothesize_device itself returns are provisional for such a call — its counts are what the launches COVER;
 * the finalize call / sc_wait deliver the real ones.) */
int sc_finalize_gathered_device_async(sc_ctx* ctx, const uint64_t* d_keys, int n_pairs, float* d_Rt, uint8_t* d_mask);

/* Phase 1 in two halves, for large T_total over several GPUs: stage B's certificate (sc_tri.hip 3b) samples ~5T/8
 * edges, which every rank would otherwise repeat (126 us at T_total = 400 k against 33 us at 50 k).  `begin` runs A,
 * the edge list and THIS rank's share of the sample (every shard_world-th sampled edge) into d_hist (device,
 * SC_HIST_WORDS x u32, zeroed here); the caller SUMS d_hist over the ranks (one 1 KiB all-reduce; any order: integer
 * sums) and passes the result to `end`, which prunes, enumerates, selects and scores exactly like
 * sc_hypothesize_device — the summed histogram counts distinct genuine triangles, so the bound it certifies is valid
 * and identical on every rank, and results equal the unsharded run's.  With shard_world == 1 the sum is the identity. */
#define SC_HIST_WORDS 256
int sc_hypothesize_begin_device(sc_ctx* ctx, const float* d_src, const float* d_tgt, int64_t n,
                                const sc_params* params, uint32_t* d_hist, sc_stats* stats);
int sc_hypothesize_end_device(sc_ctx* ctx, const uint32_t* d_hist, uint64_t* d_key, sc_stats* stats);

/* ---- stages A and B sharded too: phase API, one context per GPU / rank ------------------
 * With the two-phase form above every rank repeats stages A and B.  Here rank r computes the row block
 * [r * rows_per_rank, ...) of the adjacency bit rows (and of S, kept local, unless SC_FLAG_NO_DENSE_S) and enumerates
 * the triangles of a contiguous, equally heavy range of rows; what the ranks exchange lives in CALLER buffers, and the
 * caller runs the collectives between the phases (RCCL through torch.distributed in this repo's host layer, RCCL
 * directly in sc_register_multi below; on one GPU the "ranks" of a test simply share the buffers):
 *     sc_shard_compat_device (d_bits_all)      -> all-gather, in place, of bits_bytes_per_rank per rank
 *     sc_shard_edges_device  (d_hist)          -> all-reduce SUM of SC_HIST_WORDS u32 (1 KiB)
 *     sc_shard_select_device (d_hist, d_mine)  -> all-gather of cand_bytes_per_rank per rank into d_cand_all
 *     sc_shard_score_device  (d_cand_all, d_key) -> all-gather of the 16-byte key pairs
 *     sc_finalize_gathered_device (d_keys, shard_world, ...)
 * d_bits_all: bits_bytes_total bytes; rank r's slice starts at r * bits_bytes_per_rank (its rows at their global
 * index).  A candidate blob holds a rank's own best triangles in (i,j,k) order — up to max(2T/world, 4096) <<
 * shard_cand_level of them, never more than T (a rank contributes ~T/world with equally heavy row ranges); the row
 * ranges ascend with the rank, so the concatenated blobs are in global (i,j,k) order and the merge is the same exact
 * select + compaction every rank runs on one GPU: winner, (R,t) and mask are bit-identical to the unsharded call.  If a
 * rank had to cut its list and the cut could have mattered, the finalize call returns SC_ERETRY on every rank.  Every rank must pass the
 * same n and parameters (shard_rank aside).  shard_world <= 64; shard_world == 1 works (no collective needed).
 * Each phase leaves its last kernels queued on the context's stream; enqueue the collective on the same stream. */
typedef struct sc_shard_plan {
  uint32_t size;                 /* = sizeof(sc_shard_plan), set by the caller                              */
  uint32_t rows_per_rank;        /* rows of the adjacency matrix rank r computes: [r * rows_per_rank, ...)   */
  uint32_t words_per_row;        /* u64 words per bit row                                                   */
  uint32_t reserved;
  uint64_t bits_bytes_per_rank;  /* rows_per_rank * words_per_row * 8                                       */
  uint64_t bits_bytes_total;     /* shard_world * bits_bytes_per_rank: size of d_bits_all                    */
  uint64_t cand_bytes_per_rank;  /* size of one candidate blob; d_cand_all holds shard_world of them         */
} sc_shard_plan;
int sc_shard_plan_query(const sc_params* params, int64_t n, sc_shard_plan* plan);
int sc_shard_compat_device(sc_ctx* ctx, const float* d_src, const float* d_tgt, int64_t n, const sc_params* params,
                           void* d_bits_all);
int sc_shard_edges_device(sc_ctx* ctx, uint32_t* d_hist);
int sc_shard_select_device(sc_ctx* ctx, const uint32_t* d_hist, void* d_cand_mine);
int sc_shard_score_device(sc_ctx* ctx, const void* d_cand_all, uint64_t* d_key, sc_stats* stats);

/* ---- native multi-device entry: one process, n_dev GPUs, RCCL inside the library ------------
 * For hosts without a collective layer of their own (C++, mex): host arrays in, (R, t, mask) out, like sc_register.
 * One context per device, stages A, B and C all sharded (the phase API above), the four collectives of a call issued
 * through RCCL on the devices' streams (all-gather of the bit rows, 1 KiB all-reduce, all-gather of the candidate
 * blobs, all-gather of the key pairs).  One worker thread per device drives its GPU (a call is ~35 launches and four
 * read-backs per device: from a single thread that is ~1 ms per step for eight GPUs); the calling convention stays
 * single-caller.  RCCL is opened at run time (librccl.so.1) only when n_dev > 1: n_dev == 1 is exactly sc_register
 * and makes no RCCL call.  device_ids must be distinct.  Results are bit-identical to sc_register for every n_dev.
 * params->shard_* must be left at no sharding (rank 0, world 1).  stats: rank 0's, with tri_scored and
 * workspace_bytes summed over the devices.  Errors: the first failing rank's status; sc_multi_last_error has its text.
 * sc_create_multi_loopback: n_ranks ranks on ONE device with device copies in place of RCCL — a test hook that runs
 * the whole orchestration on a one-GPU box; with n_ranks == 1 it runs the rank machinery over a real single-rank RCCL
 * communicator instead, which executes the RCCL calls themselves (N > 1 over RCCL is unmeasured on hardware: DESIGN.md §7). */
typedef struct sc_multi sc_multi;
int         sc_create_multi(const int* device_ids, int n_dev, sc_multi** out);
int         sc_create_multi_loopback(int device, int n_ranks, sc_multi** out);
void        sc_destroy_multi(sc_multi* m);
const char* sc_multi_last_error(const sc_multi* m);
int         sc_register_multi(sc_multi* m, const float* src, const float* tgt, int64_t n, const sc_params* params,
                              float R[9], float t[3], uint8_t* mask, sc_stats* stats);

/* ---- stage-level hooks (host pointers in and out) so every kernel is parity-testable alone --------
 * All take SoA or AoS input per params->layout and run ONLY the named stage(s) on the GPU. */

/* row A: S n x n fp32 row-major, bits n x ceil(n/64) u64 (bit j%64 of word j/64 of row i), deg n u32.
 * Any output pointer may be NULL. */
int sc_compat_host(sc_ctx* ctx, const float* src, const float* tgt, int64_t n, const sc_params* params,
                   float* S, uint64_t* bits, uint32_t* deg);

/* rows A+B: ranked top-T triangles.  tri: T x 3 u32 (i<j<k, rank order), key: T u32 (fp32 bits of w for
 * SC_RANK_WEIGHT, the degree sum for SC_RANK_DEGREE), *t_eff = min(T, tri_total). */
int sc_triangles_host(sc_ctx* ctx, const float* src, const float* tgt, int64_t n, const sc_params* params,
                      uint32_t* tri, uint32_t* key, uint32_t* t_eff, uint64_t* tri_total, uint64_t* edges);

/* row C1: Rt: T x 12 fp32 (R row-major, then t) for the given triangles. */
int sc_kabsch_host(sc_ctx* ctx, const float* src, const float* tgt, int64_t n, const sc_params* params,
                   const uint32_t* tri, uint32_t n_tri, float* Rt);

/* row C2: inlier count of every hypothesis over all n correspondences, plus the arg-max key
 * (rank index = position in Rt).  cnt may be NULL. */
int sc_score_host(sc_ctx* ctx, const float* src, const float* tgt, int64_t n, const sc_params* params,
                  const float* Rt, uint32_t n_hyp, uint32_t* cnt, uint64_t* key);

/* row C3: mask of one hypothesis. */
int sc_mask_host(sc_ctx* ctx, const float* src, const float* tgt, int64_t n, const sc_params* params,
                 const float Rt[12], uint8_t* mask);

#ifdef __cplusplus
}
#endif
#endif /* SACCOT_H */
