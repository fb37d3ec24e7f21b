/*
 * saccot_debug.h — test / tuning hooks of libsaccot.so.  NOT part of the drop-in surface (include/saccot.h): a caller of
 * the registration path never needs this header.  The parity tests, the A/B sweeps under tools/ and bench.py do.
 *
 * The library reads no environment variable (SURVEY.md §5 "config / flags"): the scheduling knobs and forced fallbacks
 * live in the per-context struct below.  None of them can change a result — only launch geometry, or which of two
 * bit-identical code paths runs.  (Timing-only ablation bodies of the C2 filters, which DO return wrong counts, exist only
 * in a library built with -DSC_ABLATIONS, whose sc_debug has a field more: the product's struct has no slot for them.)
 */
#ifndef SACCOT_DEBUG_H
#define SACCOT_DEBUG_H

#include "saccot.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 = default everywhere (-1 for the two *_self_max fields).  sc_set_debug(ctx, NULL) restores the defaults.
 * Every knob selects between code paths that return IDENTICAL results (a launch geometry, a fallback the library would take by
 * itself on other inputs, a forced failure of a speculation that is then repeated); each has a parity test that sets it.  The knobs
 * of round 4's measured-and-slower variants (one-launch select, one-launch compaction, winner step inside the arg-max launch, the
 * sample inside the edge kernel, the lane = correspondence scoring kernel, the weight histogram's own launch) left with their code
 * in round 5. */
typedef struct sc_debug {
  uint32_t size;              /* = sizeof(sc_debug)                                                          */
  uint32_t no_events;         /* 1: stage B walks the bit rows twice instead of recording an event list       */
  uint64_t event_cap;         /* event records per call (>= 256): forces the overflow fallback when too small  */
  int64_t  compact_self_max;  /* key tiles up to which the compaction sums the tile counts itself (-1: 4096)  */
  int64_t  scan_self_max;     /* scan tiles up to which the down-sweep sums the block sums itself (-1: 4096)  */
  uint32_t grid_blocks[4];    /* grid sizes (0 = automatic): [0] counting pass, [1] key kernel, [2] select rounds, [3] heaviest-edge sample */
  uint32_t lanes_per_edge[4]; /* 4, 8, 16, 32, 64 (0 = default): [0] row-walking count (no 64), [1] row-walking keys, [2] certifying sample, [3] event-recording count */
  uint64_t sample_edges;      /* edges in stage B's certifying pruning sample (default ~5T/8, at least 32768)   */
  uint32_t score_split;       /* share (of 256) of the hypotheses scored by the f32-MFMA body of C2 */
  uint32_t compat_one_phase;  /* 1: stage A runs the exact chain on every pair of an interior tile            */
  uint32_t compat_rows;       /* stage A tile height: 0 = by size (16 rows below 10 000 correspondences, 32 from there), 16, 32, 64 */
  uint32_t compat_store_mode; /* stage A stores of S: 0 = by size; bit 0 = 4 bytes per lane, bit 2 = 16 bytes, bit 1 = non-temporal */
  uint32_t compat_linear_order; /* 1: stage A's 64 x 64 blocks in index order instead of the XCD-aware order   */
  uint32_t sample_mode;       /* stage B's pruning sample: 0 = chosen by entry point and size (an estimating sample where the call can be repeated, else one of the two certifying ones), 1 = every stride-th edge, 2 = the heaviest edges (1, 2: certifying) */
  uint32_t rows_unfused;      /* 1: row statistics and the scans of the row counts as separate launches             */
  uint32_t no_edge_build;     /* 1: row statistics and edge list as two launches instead of the hot path's one (launch_edge_build) */
  uint32_t no_estimate;       /* 1: stage B never prunes by an ESTIMATED bound (verified by the select, call repeated when it was too high) — always by a certifying sample, as every entry point other than sc_register / sc_register_device(_async) does anyway */
  uint32_t est_margin_pct;    /* the estimated bound aims at the key of rank (pct / 100) x T (0 = by T and rate, 115 .. 200); a small value forces the failure-and-repeat path (tests) */
  uint32_t no_fast;           /* 1: sc_register_device always waits for stage B's two counts in the middle of the call (the form every other entry point uses) instead of enqueueing the whole chain of a repeated shape host-free */
  uint32_t score_filter;      /* stage C2, inlier count: 0 = by size and scale (plain fp32 kernel for small calls; for large ones a matrix-pipe filter + exact fix-up: the Gram filter — in the frame of a voted reference hypothesis, with its triangle-inequality cut — wherever its shells fit inside tau^2, else the linear one); 1 = always plain; 2 = always the linear filter; 3 = always the Gram filter */
  uint32_t filter_splits;     /* grid.y of the filter kernel (0 = by size)                                            */
  uint32_t filter_queue_cap;  /* entries of the filter's queue of undecided tests (0 = by size): a small one forces the recount path */
  uint32_t filter_lds_queue;  /* entries of a wave's own queue, 64 .. 256 (0 = 256)                                   */
  uint32_t filter_blind;      /* 1: the host picks stage C2's kernel as if the coordinate maxima had not arrived yet (it then assumes the filter applies; the filter's own range test sends what it cannot bound to the exact recount) */
  uint32_t gram_kappa_q4;     /* the Gram filter's cut: a hypothesis counts as NEAR the call's reference frame while its reach stays under (value / 16) x tau; 0 = 8 tau (default), 1 = practically no cut (every workgroup walks every correspondence) */
  uint32_t gram_ref_late;     /* 1: the Gram filter's reference frame is voted after the selection, in a launch of its own (what every path but sc_register / sc_register_device does anyway), instead of by an extra workgroup of stage B's counting pass among the estimating sample's best triangles */
  uint32_t gram_guard_fail;   /* 1: the run-time probe of the matrix pipe's accumulation model reports a violation (tests: the Gram filter must then never be chosen) */
#ifdef SC_ABLATIONS           /* lab builds only (sac-cot_amd/build.py --ablations): NOT in the product's struct */
  uint32_t filter_variant;    /* body of the filter kernel: 1 .. 3 = bit-identical scheduling variants; >= 16 = timing-only ablations that return WRONG counts */
  uint32_t lab_pad_;
#endif
} sc_debug;
int         sc_set_debug(sc_ctx* ctx, const sc_debug* dbg);

/* Diagnostics of the LAST call on this context: which stage C2 kernel it ran and, for the matrix-pipe filter, what it
 * handed to the exact pass; how the call was enqueued.  Synchronises the context's stream. */
typedef struct sc_debug_info {
  uint32_t size;              /* = sizeof(sc_debug_info), set by the caller                                          */
  uint32_t c2_kernel;         /* 0: plain fp32 kernel (also the truncated scores); 1: linear filter + exact pass; 2: Gram filter + exact pass */
  uint64_t filter_undecided;  /* queue entries (linear: one per correspondence and wave half with >= 1 undecided test; Gram: one per correspondence, lane half and group of four hypotheses) */
  uint64_t filter_recounts;   /* (8-hypothesis wave, grid split) pairs recounted wholesale by the exact pass           */
  uint32_t filter_splits;     /* grid.y of the filter launch                                                           */
  uint32_t fast_path;         /* sc_register_device: 0 = the call waited for stage B's counts; 1 = enqueued host-free and validated at its end; 2 = enqueued host-free, failed validation (a count outgrew what the launches covered, fewer triangles than T, event overflow), repeated the waiting way */
  uint32_t gram_guard;        /* run-time probe of the matrix pipe's accumulation arithmetic (once per context, before the first call that could choose the Gram filter): 0 = not run yet, 1 = the model the Gram bound assumes holds, 2 = violated: the Gram filter is disabled for this context */
  uint32_t prune_bound;       /* stage B's pruning bound in the last call: 0 = certified by the sample (or no pruning), 1 = estimated from a 1-in-64 sample of the triangles and verified by the select, 2 = estimated, found too high by the select, call repeated with a certifying sample */
  float    gram_guard_worst;  /* largest |hardware - exact| / largest term the probe saw, in units of 2^-24 (the bound assumes 18.5) */
  uint32_t reserved2;
  /* the Gram filter's cut in the last call (c2_kernel == 2): of `gram_rows` coefficient rows (hypotheses of this rank, padded to 256)
   * `gram_near_hyp` are near the reference frame and look only at the `gram_near_corr` correspondences near it */
  uint32_t gram_near_corr, gram_near_hyp, gram_rows;
  uint32_t gram_ref;          /* position (in this rank's share of the ranked list) of the hypothesis voted reference frame; 0xFFFFFFFF: none (box-centre frame) */
  uint32_t gram_ref_votes_q8; /* its soft vote count x 256 (of 64 voters) */
  float    us_c2_filter;      /* SC_FLAG_TIMING_HOT, a filtered stage C2: the FILTER kernel's own duration, from its dispatch packet's timestamps (sc_stats.us_score spans filter + exact pass); 0 otherwise */
  /* CUMULATIVE over the context's life (r05: what a stream of frames reports — the fields above only describe the last call).
   * A frame = one sc_register / sc_register_device(_async) call, or one sc_hypothesize_device + finalize pair: */
  uint64_t n_frames;          /* frames that have come to their end                                                        */
  uint64_t n_fast_ok;         /* ... enqueued host-free and valid (fast_path 1)                                              */
  uint64_t n_fast_repeat;     /* ... enqueued host-free, void at the end (a count outgrew the cover, fewer triangles than T, event overflow, a failed estimate): repeated the waiting way (fast_path 2; SC_EBOUND on the sc_hypothesize_device path) */
  uint64_t n_est_ok;          /* ... whose estimated pruning bound the select verified (prune_bound 1)                       */
  uint64_t n_est_fail;        /* ... whose estimate was too high: repeated with a certifying sample (prune_bound 2)          */
  uint64_t cover_edges, cover_triangles;  /* what the LAST call's launches covered if it was enqueued host-free (0: it waited)  */
  uint64_t n_hostfree_grow;   /* buffers re-allocated inside host-free enqueues (each synchronises the stream: a stall in a stream of frames)  */
} sc_debug_info;
int         sc_debug_last(sc_ctx* ctx, sc_debug_info* out);

#ifdef __cplusplus
}
#endif
#endif /* SACCOT_DEBUG_H */
