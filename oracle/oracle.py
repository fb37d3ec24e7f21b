"""ctypes binding of oracle/libsaccot_oracle.so (the CPU restatement, saccot_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  PARITY UNPINNED: the reference (/root/reference/README.md:1-2) holds no
code and no vectors; see the header of saccot_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile the restatement with oracle/Makefile (gcc) if the .so is missing or stale."""
    so = os.path.join(_HERE, "libsaccot_oracle.so")
    src = os.path.join(_HERE, "saccot_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        f32p, u8p, u32p, u64p, f64p = (C.POINTER(t) for t in (C.c_float, C.c_uint8, C.c_uint32, C.c_uint64, C.c_double))
        L.so_derive.argtypes = [C.c_float, C.c_float, C.c_float, f32p]
        L.so_expf.argtypes = [C.c_float]
        L.so_expf.restype = C.c_float
        L.so_compat.argtypes = [f32p, f32p, C.c_int64, C.c_float, C.c_float, C.c_float, f32p, u64p, u32p, C.c_int]
        L.so_triangles.argtypes = [f32p, u64p, u32p, C.c_int64, C.c_int, C.c_uint32, u32p, u32p, u32p, u64p, C.c_int]
        L.so_kabsch3.argtypes = [f32p, f32p, C.c_int64, u32p, C.c_uint32, f32p, C.c_int]
        L.so_kabsch3.restype = None
        L.so_score.argtypes = [f32p, f32p, C.c_int64, f32p, C.c_uint32, C.c_float, u32p, C.c_int]
        L.so_score.restype = None
        L.so_best_key.argtypes = [u32p, C.c_uint32, u32p]
        L.so_best_key.restype = C.c_uint64
        L.so_mask.argtypes = [f32p, f32p, C.c_int64, f32p, C.c_float, u8p]
        L.so_mask.restype = None
        L.so_register.argtypes = [f32p, f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                  C.c_int, C.c_int, f32p, f32p, u8p, u64p, f64p]
        L.so_refine.argtypes = [f32p, f32p, C.c_int64, u8p, f32p]
        L.so_derive2.argtypes = [C.c_float, f32p]
        L.so_derive2.restype = None
        L.so_score_mode.argtypes = [f32p, f32p, C.c_int64, f32p, C.c_uint32, C.c_float, C.c_int, u32p, C.c_int]
        L.so_score_mode.restype = None
        L.so_register_mode.argtypes = [f32p, f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                       C.c_int, C.c_int, C.c_int, f32p, f32p, u8p, u64p, f64p]
        L.so_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a: np.ndarray | None, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def soa(points: np.ndarray) -> np.ndarray:
    """(n,3) -> contiguous 3 planes of n, float32."""
    return np.ascontiguousarray(np.asarray(points, dtype=np.float32).T)


def derive(sigma: float, t_cmp: float, tau: float) -> tuple[float, float, float]:
    out = np.zeros(3, dtype=np.float32)
    if lib().so_derive(sigma, t_cmp, tau, _p(out, C.c_float)) != 0:
        raise ValueError("bad parameters")
    return float(out[0]), float(out[1]), float(out[2])


def expf(x: float) -> float:
    return float(lib().so_expf(np.float32(x)))


def max_threads() -> int:
    return int(lib().so_max_threads())


def compat(src: np.ndarray, tgt: np.ndarray, sigma: float, t_cmp: float, min_len: float, tau: float = 1.0,
           threads: int = 1, want_S: bool = True):
    """Stage A.  Returns S (n,n) f32 | None, bits (n,W) u64, deg (n,) u32."""
    n = src.shape[0]
    d_thr, nis, _ = derive(sigma, t_cmp, tau)
    ps, qs = soa(src), soa(tgt)
    W = (n + 63) // 64
    S = np.empty((n, n), dtype=np.float32) if want_S else None
    bits = np.zeros((n, W), dtype=np.uint64)
    deg = np.zeros(n, dtype=np.uint32)
    rc = lib().so_compat(_p(ps, C.c_float), _p(qs, C.c_float), n, d_thr, np.float32(min_len), nis,
                         _p(S, C.c_float), _p(bits, C.c_uint64), _p(deg, C.c_uint32), threads)
    if rc != 0:
        raise RuntimeError(f"so_compat rc={rc}")
    return S, bits, deg


def triangles(S: np.ndarray | None, bits: np.ndarray, deg: np.ndarray, T: int, rank_mode: int = 0, threads: int = 1):
    """Stage B.  Returns tri (t_eff,3) u32, key (t_eff,) u32, tri_total."""
    n = bits.shape[0]
    tri = np.zeros((max(T, 1), 3), dtype=np.uint32)
    key = np.zeros(max(T, 1), dtype=np.uint32)
    t_eff = C.c_uint32(0)
    total = C.c_uint64(0)
    rc = lib().so_triangles(_p(S, C.c_float), _p(bits, C.c_uint64), _p(deg, C.c_uint32), n, rank_mode, T,
                            _p(tri, C.c_uint32), _p(key, C.c_uint32), C.byref(t_eff), C.byref(total), threads)
    if rc != 0:
        raise RuntimeError(f"so_triangles rc={rc}")
    return tri[: t_eff.value].copy(), key[: t_eff.value].copy(), int(total.value)


def kabsch3(src: np.ndarray, tgt: np.ndarray, tri: np.ndarray, threads: int = 1) -> np.ndarray:
    """Stage C1.  Returns Rt (T,12) f32."""
    n = src.shape[0]
    tri = np.ascontiguousarray(tri, dtype=np.uint32)
    Rt = np.zeros((tri.shape[0], 12), dtype=np.float32)
    ps, qs = soa(src), soa(tgt)
    lib().so_kabsch3(_p(ps, C.c_float), _p(qs, C.c_float), n, _p(tri, C.c_uint32), tri.shape[0], _p(Rt, C.c_float), threads)
    return Rt


def score(src: np.ndarray, tgt: np.ndarray, Rt: np.ndarray, tau: float, threads: int = 1, score_mode: int = 0) -> np.ndarray:
    """Stage C2.  Returns cnt (T,) u32: inlier counts, or the truncated scores of score_mode 1 / 2."""
    n = src.shape[0]
    Rt = np.ascontiguousarray(Rt, dtype=np.float32)
    cnt = np.zeros(Rt.shape[0], dtype=np.uint32)
    tau2 = np.float32(np.float64(np.float32(tau)) * np.float64(np.float32(tau)))  # tau is an fp32 parameter (sc_params.tau): round first
    ps, qs = soa(src), soa(tgt)
    if score_mode:
        inv = np.zeros(2, dtype=np.float32)
        lib().so_derive2(np.float32(tau), _p(inv, C.c_float))
        lib().so_score_mode(_p(ps, C.c_float), _p(qs, C.c_float), n, _p(Rt, C.c_float), Rt.shape[0],
                            inv[score_mode - 1], score_mode, _p(cnt, C.c_uint32), threads)
        return cnt
    lib().so_score(_p(ps, C.c_float), _p(qs, C.c_float), n, _p(Rt, C.c_float), Rt.shape[0], tau2, _p(cnt, C.c_uint32), threads)
    return cnt


def best_key(cnt: np.ndarray, rank_of: np.ndarray | None = None) -> int:
    cnt = np.ascontiguousarray(cnt, dtype=np.uint32)
    ro = None if rank_of is None else np.ascontiguousarray(rank_of, dtype=np.uint32)
    return int(lib().so_best_key(_p(cnt, C.c_uint32), cnt.shape[0], _p(ro, C.c_uint32)))


def mask(src: np.ndarray, tgt: np.ndarray, Rt12: np.ndarray, tau: float) -> np.ndarray:
    n = src.shape[0]
    Rt12 = np.ascontiguousarray(Rt12, dtype=np.float32)
    out = np.zeros(n, dtype=np.uint8)
    tau2 = np.float32(np.float64(np.float32(tau)) * np.float64(np.float32(tau)))  # tau is an fp32 parameter (sc_params.tau): round first
    ps, qs = soa(src), soa(tgt)
    lib().so_mask(_p(ps, C.c_float), _p(qs, C.c_float), n, _p(Rt12, C.c_float), tau2, _p(out, C.c_uint8))
    return out


def register(src: np.ndarray, tgt: np.ndarray, sigma: float, t_cmp: float, tau: float, min_len: float,
             max_triangles: int, rank_mode: int = 0, threads: int = 1, score_mode: int = 0):
    """Whole path.  Returns dict(rc, R, t, mask, edges, tri_total, t_eff, best_rank, best_count, stage_s)."""
    n = src.shape[0]
    ps, qs = soa(src), soa(tgt)
    R = np.zeros(9, dtype=np.float32)
    t = np.zeros(3, dtype=np.float32)
    m = np.zeros(n, dtype=np.uint8)
    st = np.zeros(5, dtype=np.uint64)
    ss = np.zeros(5, dtype=np.float64)
    rc = lib().so_register_mode(_p(ps, C.c_float), _p(qs, C.c_float), n, sigma, t_cmp, tau, min_len, max_triangles,
                                rank_mode, score_mode, threads, _p(R, C.c_float), _p(t, C.c_float), _p(m, C.c_uint8),
                                _p(st, C.c_uint64), _p(ss, C.c_double))
    return dict(rc=rc, R=R.reshape(3, 3), t=t, mask=m, edges=int(st[0]), tri_total=int(st[1]), t_eff=int(st[2]),
                best_rank=int(st[3]), best_count=int(st[4]), stage_s=ss)


def refine(src: np.ndarray, tgt: np.ndarray, mask_: np.ndarray, Rt12: np.ndarray):
    """fp64 least-squares refit over the inlier mask.  Returns (done, Rt12 float32)."""
    n = src.shape[0]
    Rt = np.ascontiguousarray(Rt12, dtype=np.float32).reshape(12).copy()
    mk = np.ascontiguousarray(mask_, dtype=np.uint8)
    ps, qs = soa(src), soa(tgt)
    done = lib().so_refine(_p(ps, C.c_float), _p(qs, C.c_float), n, _p(mk, C.c_uint8), _p(Rt, C.c_float))
    return bool(done), Rt
