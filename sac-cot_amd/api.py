"""Host-side mirror of the C ABI in include/saccot.h (ctypes over libsaccot.so).

The reference has no operator/plugin interface to mirror (/root/reference/README.md:1-2 is the whole tree), so the
names follow SURVEY.md §8(b): `register()` is the drop-in entry point — correspondences in, (R, t, inlier mask)
out — and the `compat` / `triangles` / `kabsch` / `score` / `mask` methods are the per-stage hooks the parity
tests drive.  Everything computes on the GPU through libsaccot.so; if the library or a HIP device is missing
this module raises — there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsaccot.so")

SC_OK, SC_EINVAL, SC_ENOMEM, SC_EHIP, SC_ERCCL, SC_ENOHYP, SC_ETOOMANY, SC_ERETRY, SC_EBOUND = 0, -1, -2, -3, -4, -5, -6, -7, -8
SC_FLAG_SHARD_AB = 4096  # sc_register_multi: shard stages A and B at every size (default: replicated below 8192 correspondences)
SC_FLAG_EST_BOUND = 128  # phase API sc_shard_*: prune by an estimated bound, no histogram all-reduce, SC_EBOUND -> repeat without it
SC_AOS, SC_SOA = 0, 1
SC_RANK_WEIGHT, SC_RANK_DEGREE = 0, 1
SC_SCORE_COUNT, SC_SCORE_MSE, SC_SCORE_MAE = 0, 1, 2
SC_FLAG_TIMING, SC_FLAG_EXACT_TOTAL, SC_FLAG_NO_PRUNE, SC_FLAG_REFINE, SC_FLAG_TIMING_HOT, SC_FLAG_NO_DENSE_S = 1, 2, 4, 8, 16, 32
SC_FLAG_TIMING_ONE = 64


def SC_TIMING_STAGE(k: int) -> int:
    """flags bits selecting the one bracket of SC_FLAG_TIMING_ONE: 0 staging, 1 compat, 2 triangles, 3 kabsch, 4 score,
    5 argmax, 6 mask"""
    return (k & 15) << 8


SC_HIST_WORDS = 256  # u32 words of the pruning-sample histogram (sc_hypothesize_begin_device)

EXPORTS = ["sc_version", "sc_strerror", "sc_default_params", "sc_create", "sc_destroy", "sc_set_stream",
           "sc_last_error", "sc_set_debug", "sc_debug_last", "sc_register", "sc_register_device", "sc_register_device_async", "sc_wait",
           "sc_hypothesize_device", "sc_finalize_device",
           "sc_hypothesize_begin_device", "sc_hypothesize_end_device", "sc_finalize_gathered_device", "sc_finalize_gathered_device_async",
           "sc_shard_plan_query", "sc_shard_compat_device", "sc_shard_edges_device", "sc_shard_select_device",
           "sc_shard_score_device", "sc_create_multi", "sc_create_multi_loopback", "sc_destroy_multi",
           "sc_multi_last_error", "sc_register_multi",
           "sc_compat_host", "sc_triangles_host", "sc_kabsch_host", "sc_score_host", "sc_mask_host"]


class ScParams(C.Structure):
    _fields_ = [("size", C.c_uint32), ("sigma", C.c_float), ("t_cmp", C.c_float), ("tau", C.c_float),
                ("min_len", C.c_float), ("max_triangles", C.c_uint32), ("rank_mode", C.c_int32),
                ("layout", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
                ("shard_block", C.c_uint32), ("flags", C.c_uint32), ("max_workspace", C.c_uint64),
                ("score_mode", C.c_int32), ("shard_cand_level", C.c_int32)]


class ScStats(C.Structure):
    _fields_ = [("size", C.c_uint32), ("n", C.c_uint32), ("edges", C.c_uint64), ("tri_total", C.c_uint64),
                ("tri_kept", C.c_uint32), ("tri_scored", C.c_uint32), ("best_rank", C.c_uint32),
                ("best_count", C.c_uint32), ("us_stage", C.c_float), ("us_compat", C.c_float), ("us_triangles", C.c_float),
                ("us_trikeys", C.c_float),
                ("us_kabsch", C.c_float), ("us_score", C.c_float), ("us_argmax", C.c_float), ("us_mask", C.c_float),
                ("us_total", C.c_float),
                ("workspace_bytes", C.c_uint64), ("bytes_moved", C.c_uint64)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "size"}


class ScShardPlan(C.Structure):
    """Mirror of `sc_shard_plan` (include/saccot.h): sizes of the buffers the ranks exchange when A and B are sharded."""
    _fields_ = [("size", C.c_uint32), ("rows_per_rank", C.c_uint32), ("words_per_row", C.c_uint32),
                ("reserved", C.c_uint32), ("bits_bytes_per_rank", C.c_uint64), ("bits_bytes_total", C.c_uint64),
                ("cand_bytes_per_rank", C.c_uint64)]


class ScDebug(C.Structure):
    """Mirror of `sc_debug` (include/saccot_debug.h): test / tuning hook, 0 = default (-1 for the *_self_max fields).  The two
    arrays are set through the names of their slots (`_ALIASES`): set_debug(tg_count=4, cnt_blocks=37)."""
    _fields_ = [("size", C.c_uint32), ("no_events", C.c_uint32), ("event_cap", C.c_uint64),
                ("compact_self_max", C.c_int64), ("scan_self_max", C.c_int64),
                ("grid_blocks", C.c_uint32 * 4), ("lanes_per_edge", C.c_uint32 * 4),
                ("sample_edges", C.c_uint64), ("score_split", C.c_uint32), ("compat_one_phase", C.c_uint32),
                ("compat_rows", C.c_uint32), ("compat_store_mode", C.c_uint32), ("compat_linear_order", C.c_uint32),
                ("sample_mode", C.c_uint32), ("rows_unfused", C.c_uint32), ("no_edge_build", C.c_uint32),
                ("no_estimate", C.c_uint32), ("est_margin_pct", C.c_uint32), ("no_fast", C.c_uint32),
                ("score_filter", C.c_uint32), ("filter_splits", C.c_uint32), ("filter_queue_cap", C.c_uint32),
                ("filter_lds_queue", C.c_uint32), ("filter_blind", C.c_uint32), ("gram_kappa_q4", C.c_uint32),
                ("gram_ref_late", C.c_uint32), ("gram_guard_fail", C.c_uint32)]
    _ALIASES = {"cnt_blocks": ("grid_blocks", 0), "keys_blocks": ("grid_blocks", 1), "sel_blocks": ("grid_blocks", 2),
                "sample_blocks": ("grid_blocks", 3), "tg_count": ("lanes_per_edge", 0), "tg_keys": ("lanes_per_edge", 1),
                "tg_sample": ("lanes_per_edge", 2), "tg_events": ("lanes_per_edge", 3)}


class ScDebugLab(C.Structure):
    """`sc_debug` of a library built with -DSC_ABLATIONS (sac-cot_amd/build.py --ablations): one knob more."""
    _fields_ = ScDebug._fields_ + [("filter_variant", C.c_uint32), ("lab_pad_", C.c_uint32)]
    _ALIASES = ScDebug._ALIASES


class ScDebugInfo(C.Structure):
    """Mirror of `sc_debug_info` (include/saccot_debug.h): which C2 kernel the last call ran, the filter's hand-overs,
    how the call was enqueued (fast_path: 0 waited, 1 host-free, 2 host-free then repeated), the matrix-pipe probe."""
    _fields_ = [("size", C.c_uint32), ("c2_kernel", C.c_uint32), ("filter_undecided", C.c_uint64),
                ("filter_recounts", C.c_uint64), ("filter_splits", C.c_uint32), ("fast_path", C.c_uint32),
                ("gram_guard", C.c_uint32), ("prune_bound", C.c_uint32), ("gram_guard_worst", C.c_float),
                ("reserved2", C.c_uint32), ("gram_near_corr", C.c_uint32), ("gram_near_hyp", C.c_uint32),
                ("gram_rows", C.c_uint32), ("gram_ref", C.c_uint32), ("gram_ref_votes_q8", C.c_uint32),
                ("us_c2_filter", C.c_float),
                ("n_frames", C.c_uint64), ("n_fast_ok", C.c_uint64), ("n_fast_repeat", C.c_uint64),
                ("n_est_ok", C.c_uint64), ("n_est_fail", C.c_uint64), ("cover_edges", C.c_uint64), ("cover_triangles", C.c_uint64), ("n_hostfree_grow", C.c_uint64)]


class SacCotError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libsaccot status {status}: {msg}")
        self.status = status


_LIB = None


def load_library() -> C.CDLL:
    """dlopen libsaccot.so and declare every prototype.  Raises if it was not built (run __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's), and the
    # copy that is loaded first serves every later DT_NEEDED.  torch cannot run on the system copy ("No HIP GPUs
    # are available"), libsaccot.so runs on either — so when torch is present, let it load first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, f32p, u8p, u32p, u64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    pp, sp = C.POINTER(ScParams), C.POINTER(ScStats)
    L.sc_version.restype = C.c_int
    L.sc_strerror.argtypes = [C.c_int]; L.sc_strerror.restype = C.c_char_p
    L.sc_default_params.argtypes = [pp]; L.sc_default_params.restype = None
    L.sc_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.sc_destroy.argtypes = [vp]; L.sc_destroy.restype = None
    L.sc_set_stream.argtypes = [vp, vp]
    L.sc_last_error.argtypes = [vp]; L.sc_last_error.restype = C.c_char_p
    L.sc_set_debug.argtypes = [vp, C.POINTER(ScDebug)]
    L.sc_debug_last.argtypes = [vp, C.POINTER(ScDebugInfo)]
    L.sc_register.argtypes = [vp, f32p, f32p, C.c_int64, pp, f32p, f32p, u8p, sp]
    L.sc_register_device.argtypes = [vp, vp, vp, C.c_int64, pp, vp, vp, sp]
    L.sc_register_device_async.argtypes = [vp, vp, vp, C.c_int64, pp, vp, vp]
    L.sc_wait.argtypes = [vp, sp]
    L.sc_hypothesize_device.argtypes = [vp, vp, vp, C.c_int64, pp, vp, sp]
    L.sc_finalize_device.argtypes = [vp, vp, vp, vp, sp]
    L.sc_hypothesize_begin_device.argtypes = [vp, vp, vp, C.c_int64, pp, vp, sp]
    L.sc_hypothesize_end_device.argtypes = [vp, vp, vp, sp]
    L.sc_finalize_gathered_device.argtypes = [vp, vp, C.c_int, vp, vp, sp]
    L.sc_finalize_gathered_device_async.argtypes = [vp, vp, C.c_int, vp, vp]
    L.sc_shard_plan_query.argtypes = [pp, C.c_int64, C.POINTER(ScShardPlan)]
    L.sc_shard_compat_device.argtypes = [vp, vp, vp, C.c_int64, pp, vp]
    L.sc_shard_edges_device.argtypes = [vp, vp]
    L.sc_shard_select_device.argtypes = [vp, vp, vp]
    L.sc_shard_score_device.argtypes = [vp, vp, vp, sp]
    L.sc_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.sc_create_multi_loopback.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.sc_destroy_multi.argtypes = [vp]; L.sc_destroy_multi.restype = None
    L.sc_multi_last_error.argtypes = [vp]; L.sc_multi_last_error.restype = C.c_char_p
    L.sc_register_multi.argtypes = [vp, f32p, f32p, C.c_int64, pp, f32p, f32p, u8p, sp]
    L.sc_compat_host.argtypes = [vp, f32p, f32p, C.c_int64, pp, f32p, u64p, u32p]
    L.sc_triangles_host.argtypes = [vp, f32p, f32p, C.c_int64, pp, u32p, u32p, u32p, u64p, u64p]
    L.sc_kabsch_host.argtypes = [vp, f32p, f32p, C.c_int64, pp, u32p, C.c_uint32, f32p]
    L.sc_score_host.argtypes = [vp, f32p, f32p, C.c_int64, pp, f32p, C.c_uint32, u32p, u64p]
    L.sc_mask_host.argtypes = [vp, f32p, f32p, C.c_int64, pp, f32p, u8p]
    _LIB = L
    return L


def make_params(sigma=0.1, t_cmp=0.9, tau=0.1, min_len=0.1, max_triangles=50000, rank_mode=SC_RANK_WEIGHT,
                layout=SC_AOS, shard_rank=0, shard_world=1, shard_block=1024, flags=0, max_workspace=0,
                score_mode=0, shard_cand_level=0) -> ScParams:
    return ScParams(C.sizeof(ScParams), sigma, t_cmp, tau, min_len, max_triangles, rank_mode, layout, shard_rank,
                    shard_world, shard_block, flags, max_workspace, score_mode, shard_cand_level)


def shard_plan(params: ScParams, n: int) -> ScShardPlan:
    """sc_shard_plan_query: sizes of d_bits_all / the candidate blobs for `params` (shard_world) and n correspondences."""
    plan = ScShardPlan(size=C.sizeof(ScShardPlan))
    rc = load_library().sc_shard_plan_query(C.byref(params), n, C.byref(plan))
    if rc != SC_OK:
        raise SacCotError(rc, "sc_shard_plan_query")
    return plan


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Registrar:
    """One GPU context (`sc_ctx`): one process, one GPU, one stream."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.sc_create(device, C.byref(h))
        if rc != SC_OK:
            raise SacCotError(rc, "sc_create failed (no usable HIP device? this library has no CPU fallback): "
                              + self._lib.sc_strerror(rc).decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, allow=()):
        if rc != SC_OK and rc not in allow:
            raise SacCotError(rc, self._lib.sc_strerror(rc).decode() + " — " + self._lib.sc_last_error(self._h).decode())
        return rc

    def set_stream(self, stream_ptr: int | None):
        """Enqueue on a caller stream: pass `torch.cuda.current_stream().cuda_stream` (0 = the default stream, mapped
        to SC_STREAM_DEFAULT here).  None restores the context's private, NON-blocking stream — nothing the caller
        enqueues elsewhere (an all-reduce of the key, a copy of the mask) is ordered against it."""
        if stream_ptr is None:
            ptr = 0
        elif stream_ptr == 0:
            ptr = 1  # SC_STREAM_DEFAULT
        else:
            ptr = stream_ptr
        self._check(self._lib.sc_set_stream(self._h, C.c_void_p(ptr)))

    def set_debug(self, **knobs):
        """sc_set_debug: scheduling knobs / forced fallbacks for tests and sweeps (field names of `sc_debug`); no
        arguments restores the defaults.  The library itself reads no environment variable."""
        if not knobs:
            self._check(self._lib.sc_set_debug(self._h, None))
            return
        cls = ScDebugLab if "filter_variant" in knobs else ScDebug   # (lab builds have a field more; the product rejects the lab struct's size)
        d = cls(size=C.sizeof(cls), compact_self_max=-1, scan_self_max=-1)
        names = dict(cls._fields_)
        for k, v in knobs.items():
            if k in cls._ALIASES:
                arr, slot = cls._ALIASES[k]
                getattr(d, arr)[slot] = int(v)
            elif k in names and k not in ("size", "grid_blocks", "lanes_per_edge", "lab_pad_"):
                setattr(d, k, int(v))
            else:
                raise KeyError(f"sc_debug has no field {k!r}")
        self._check(self._lib.sc_set_debug(self._h, C.cast(C.byref(d), C.POINTER(ScDebug))))

    def debug_last(self) -> dict:
        """sc_debug_last: which stage C2 kernel the last call ran (0 plain fp32, 1 linear filter + exact pass, 2 Gram
        filter + exact pass), what the filter handed to the exact pass, how the call was enqueued (fast_path) and the result
        of the matrix-pipe probe (gram_guard).  Synchronises the context's stream."""
        d = ScDebugInfo(size=C.sizeof(ScDebugInfo))
        self._check(self._lib.sc_debug_last(self._h, C.byref(d)))
        return {k: getattr(d, k) for k, _ in ScDebugInfo._fields_ if k not in ("size", "reserved", "reserved2")}

    # ---- drop-in entry point ----------------------------------------------------------------------------
    def register(self, src, tgt, params: ScParams | None = None, **kw):
        """(n,3) src/tgt correspondences -> dict(status, R (3,3), t (3,), mask (n,) uint8, stats)."""
        p = params or make_params(**kw)
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if p.layout == SC_AOS else src.shape[1]
        R = np.zeros(9, np.float32); t = np.zeros(3, np.float32); mask = np.zeros(n, np.uint8)
        st = ScStats(C.sizeof(ScStats))
        rc = self._check(self._lib.sc_register(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(p),
                                               _p(R, C.c_float), _p(t, C.c_float), _p(mask, C.c_uint8), C.byref(st)),
                         allow=(SC_ENOHYP,))
        return dict(status=rc, R=R.reshape(3, 3), t=t, mask=mask, stats=st.as_dict())

    # ---- device-resident forms (pointers are ints: torch .data_ptr()) -------------------------------------
    def register_device(self, d_src: int, d_tgt: int, n: int, params: ScParams, d_Rt: int, d_mask: int):
        st = ScStats(C.sizeof(ScStats))
        rc = self._check(self._lib.sc_register_device(self._h, d_src, d_tgt, n, C.byref(params), d_Rt, d_mask,
                                                      C.byref(st)), allow=(SC_ENOHYP,))
        return rc, st.as_dict()

    def register_device_async(self, d_src: int, d_tgt: int, n: int, params: ScParams, d_Rt: int, d_mask: int):
        """sc_register_device_async: enqueue the whole path and return; `wait()` delivers status and statistics.  At most
        one call outstanding per Registrar; inputs must stay valid until wait() returns."""
        self._check(self._lib.sc_register_device_async(self._h, d_src, d_tgt, n, C.byref(params), d_Rt, d_mask))

    def wait(self):
        """sc_wait: the second half of register_device_async / finalize_gathered_device_async."""
        st = ScStats(C.sizeof(ScStats))
        rc = self._check(self._lib.sc_wait(self._h, C.byref(st)), allow=(SC_ENOHYP, SC_ERETRY, SC_EBOUND))
        return rc, st.as_dict()

    def hypothesize_device(self, d_src: int, d_tgt: int, n: int, params: ScParams, d_key: int):
        st = ScStats(C.sizeof(ScStats))
        self._check(self._lib.sc_hypothesize_device(self._h, d_src, d_tgt, n, C.byref(params), d_key, C.byref(st)))
        return st.as_dict()

    def hypothesize_begin_device(self, d_src: int, d_tgt: int, n: int, params: ScParams, d_hist: int):
        """Phase 1, first half (include/saccot.h): A, edges, and this rank's share of the pruning sample into
        d_hist (SC_HIST_WORDS u32 on the device, zeroed by the call).  Sum d_hist over the ranks
        (shard.allreduce_hist), then call hypothesize_end_device."""
        st = ScStats(C.sizeof(ScStats))
        self._check(self._lib.sc_hypothesize_begin_device(self._h, d_src, d_tgt, n, C.byref(params), d_hist, C.byref(st)))
        return st.as_dict()

    def hypothesize_end_device(self, d_hist: int, d_key: int):
        st = ScStats(C.sizeof(ScStats))
        self._check(self._lib.sc_hypothesize_end_device(self._h, d_hist, d_key, C.byref(st)))
        return st.as_dict()

    def finalize_device(self, d_key: int, d_Rt: int, d_mask: int):
        st = ScStats(C.sizeof(ScStats))
        rc = self._check(self._lib.sc_finalize_device(self._h, d_key, d_Rt, d_mask, C.byref(st)), allow=(SC_ENOHYP,))
        return rc, st.as_dict()

    def finalize_gathered_device(self, d_keys: int, n_pairs: int, d_Rt: int, d_mask: int):
        """Phase 2 on n_pairs all-gathered key pairs (shard.allgather_best): the reduction runs in the kernel."""
        st = ScStats(C.sizeof(ScStats))
        rc = self._check(self._lib.sc_finalize_gathered_device(self._h, d_keys, n_pairs, d_Rt, d_mask, C.byref(st)),
                         allow=(SC_ENOHYP, SC_ERETRY, SC_EBOUND))  # SC_ERETRY (sharded A + B): repeat with shard_cand_level + 1; SC_EBOUND: repeat without SC_FLAG_EST_BOUND
        return rc, st.as_dict()

    def finalize_gathered_device_async(self, d_keys: int, n_pairs: int, d_Rt: int, d_mask: int):
        """sc_finalize_gathered_device_async: the finalize kernel is enqueued; `wait()` delivers status and statistics."""
        self._check(self._lib.sc_finalize_gathered_device_async(self._h, d_keys, n_pairs, d_Rt, d_mask))

    # ---- stages A and B sharded too (include/saccot.h "phase API") ----------------------------
    def shard_compat_device(self, d_src: int, d_tgt: int, n: int, params: ScParams, d_bits_all: int):
        """Phase 1: this rank's row block of the adjacency bit rows into the shared d_bits_all -> all-gather in place."""
        self._check(self._lib.sc_shard_compat_device(self._h, d_src, d_tgt, n, C.byref(params), d_bits_all))

    def shard_edges_device(self, d_hist: int):
        """Phase 2 (bit rows gathered): degrees, edge list, this rank's share of the pruning sample -> all-reduce SUM."""
        self._check(self._lib.sc_shard_edges_device(self._h, d_hist))

    def shard_select_device(self, d_hist: int, d_cand_mine: int):
        """Phase 3: this rank's own top-T triangles (its contiguous row range) into its candidate blob -> all-gather."""
        self._check(self._lib.sc_shard_select_device(self._h, d_hist, d_cand_mine))

    def shard_score_device(self, d_cand_all: int, d_key: int):
        """Phase 4: merge of the gathered blobs, then C1 + C2 on this rank's blocks -> key pair -> all-gather, finalize."""
        st = ScStats(C.sizeof(ScStats))
        self._check(self._lib.sc_shard_score_device(self._h, d_cand_all, d_key, C.byref(st)))
        return st.as_dict()

    # ---- stage hooks -----------------------------------------------------------------------------------------
    def compat(self, src, tgt, params: ScParams, want_S=True):
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if params.layout == SC_AOS else src.shape[1]
        W = (n + 63) // 64
        S = np.empty((n, n), np.float32) if want_S else None
        bits = np.zeros((n, W), np.uint64); deg = np.zeros(n, np.uint32)
        self._check(self._lib.sc_compat_host(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(params),
                                             _p(S, C.c_float), _p(bits, C.c_uint64), _p(deg, C.c_uint32)))
        return S, bits, deg

    def triangles(self, src, tgt, params: ScParams):
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if params.layout == SC_AOS else src.shape[1]
        T = params.max_triangles
        tri = np.zeros((T, 3), np.uint32); key = np.zeros(T, np.uint32)
        t_eff = C.c_uint32(0); total = C.c_uint64(0); edges = C.c_uint64(0)
        self._check(self._lib.sc_triangles_host(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(params),
                                                _p(tri, C.c_uint32), _p(key, C.c_uint32), C.byref(t_eff),
                                                C.byref(total), C.byref(edges)))
        return tri[: t_eff.value].copy(), key[: t_eff.value].copy(), int(total.value), int(edges.value)

    def kabsch(self, src, tgt, params: ScParams, tri):
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if params.layout == SC_AOS else src.shape[1]
        tri = np.ascontiguousarray(tri, dtype=np.uint32)
        Rt = np.zeros((tri.shape[0], 12), np.float32)
        self._check(self._lib.sc_kabsch_host(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(params),
                                             _p(tri, C.c_uint32), tri.shape[0], _p(Rt, C.c_float)))
        return Rt

    def score(self, src, tgt, params: ScParams, Rt):
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if params.layout == SC_AOS else src.shape[1]
        Rt = _f32c(Rt)
        cnt = np.zeros(Rt.shape[0], np.uint32); key = C.c_uint64(0)
        self._check(self._lib.sc_score_host(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(params),
                                            _p(Rt, C.c_float), Rt.shape[0], _p(cnt, C.c_uint32), C.byref(key)))
        return cnt, int(key.value)

    def mask(self, src, tgt, params: ScParams, Rt12):
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if params.layout == SC_AOS else src.shape[1]
        Rt12 = _f32c(Rt12).reshape(12)
        m = np.zeros(n, np.uint8)
        self._check(self._lib.sc_mask_host(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(params),
                                           _p(Rt12, C.c_float), _p(m, C.c_uint8)))
        return m


class MultiRegistrar:
    """sc_multi (include/saccot.h): one process, several GPUs, RCCL inside the library.  `devices`: distinct device
    ids; `loopback_ranks` > 0 instead runs that many ranks on devices[0] with device copies in place of RCCL (test
    hook: the whole orchestration on a one-GPU box)."""

    def __init__(self, devices=(0,), loopback_ranks: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        if loopback_ranks:
            rc = self._lib.sc_create_multi_loopback(int(devices[0]), loopback_ranks, C.byref(h))
        else:
            ids = (C.c_int * len(devices))(*devices)
            rc = self._lib.sc_create_multi(ids, len(devices), C.byref(h))
        if rc != SC_OK:
            raise SacCotError(rc, "sc_create_multi failed: " + self._lib.sc_strerror(rc).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sc_destroy_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def register(self, src, tgt, params: ScParams | None = None, **kw):
        p = params or make_params(**kw)
        src, tgt = _f32c(src), _f32c(tgt)
        n = src.shape[0] if p.layout == SC_AOS else src.shape[1]
        R = np.zeros(9, np.float32); t = np.zeros(3, np.float32); mask = np.zeros(n, np.uint8)
        st = ScStats(C.sizeof(ScStats))
        rc = self._lib.sc_register_multi(self._h, _p(src, C.c_float), _p(tgt, C.c_float), n, C.byref(p),
                                         _p(R, C.c_float), _p(t, C.c_float), _p(mask, C.c_uint8), C.byref(st))
        if rc not in (SC_OK, SC_ENOHYP):
            raise SacCotError(rc, self._lib.sc_strerror(rc).decode() + " — " + self._lib.sc_multi_last_error(self._h).decode())
        return dict(status=rc, R=R.reshape(3, 3), t=t, mask=mask, stats=st.as_dict())


def register(src, tgt, device: int = 0, **kw):
    """One-shot convenience: create a context, run the whole path, destroy the context."""
    r = Registrar(device)
    try:
        return r.register(src, tgt, **kw)
    finally:
        r.close()
