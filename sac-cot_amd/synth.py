"""Synthetic correspondence sets of the sizes BASELINE.json names (SURVEY.md §8d "Synthetic inputs").

The reference ships no data (its tree is /root/reference/README.md:1-2), and BASELINE.json asks for
"synthetic correspondence sets of the named sizes", so the five configs are shape templates only.

Everything is a pure function of (seed, stream, index) through a splitmix64 counter hash, evaluated in
uint64 / float64 numpy with only + - * / sqrt (no libm transcendental: the Gaussian is Irwin-Hall(12)),
so the same arrays come out on every machine, then cast once to float32.
"""
from __future__ import annotations

import dataclasses

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output step on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _bits(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    key = splitmix64(np.array([(seed ^ (stream << 56)) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        return splitmix64(key + idx.astype(np.uint64))


def uniform01(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """float64 uniforms in [0,1) with 53 random bits."""
    return (_bits(seed, stream, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gauss(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """Approximately N(0,1): sum of 12 uniforms minus 6 (exact arithmetic order, no libm)."""
    idx = np.asarray(idx, dtype=np.uint64)
    acc = np.zeros(idx.shape, dtype=np.float64)
    for k in range(12):
        acc = acc + uniform01(seed, stream, idx * np.uint64(12) + np.uint64(k))
    return acc - 6.0


@dataclasses.dataclass(frozen=True)
class Config:
    """One BASELINE.json config as a shape template."""
    name: str
    n: int            # correspondences
    rho: float        # inlier ratio
    L: float          # scene extent (source points uniform in [-L/2, L/2]^3)
    tau: float        # inlier distance; noise sigma = tau/3
    T: int            # ranked triangles scored
    gpus: int
    seed: int

    def params(self) -> dict:
        """Path parameters per SURVEY §8d: sigma = tau, t_cmp = 0.90, min_len = tau, weight ranking."""
        return dict(sigma=self.tau, t_cmp=0.90, tau=self.tau, min_len=self.tau, max_triangles=self.T,
                    rank_mode=0)


# BASELINE.json `configs`[0..4] -> C0..C4 (SURVEY §8a table, §8d parameters)
CONFIGS = {
    "C0": Config("C0", 500, 0.30, 1.0, 0.05, 200, 1, 1000),
    "C1": Config("C1", 2000, 0.20, 1.0, 0.02, 10_000, 1, 1001),
    "C2": Config("C2", 5000, 0.15, 3.0, 0.10, 50_000, 1, 1002),
    "C3": Config("C3", 20000, 0.10, 50.0, 0.60, 200_000, 8, 1003),
    "C4": Config("C4", 5000, 0.10, 3.0, 0.10, 500_000, 8, 1004),
}


@dataclasses.dataclass
class Scene:
    src: np.ndarray       # (n,3) float32
    tgt: np.ndarray       # (n,3) float32
    R_gt: np.ndarray      # (3,3) float64
    t_gt: np.ndarray      # (3,) float64
    inlier: np.ndarray    # (n,) bool — true correspondences


def make_scene(n: int, rho: float, L: float, tau: float, seed: int) -> Scene:
    m = np.arange(n, dtype=np.uint64)
    c3 = np.arange(3, dtype=np.uint64)
    idx3 = m[:, None] * np.uint64(3) + c3[None, :]
    p = (uniform01(seed, 1, idx3) - 0.5) * L
    qn = gauss(seed, 2, np.arange(4, dtype=np.uint64))
    qn = qn / np.sqrt((qn * qn).sum())
    w, x, y, z = qn
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)
    t = (uniform01(seed, 3, c3) - 0.5) * L
    # q_true = R p + t, written out (no BLAS) so the summation order is fixed
    qt = np.empty_like(p)
    for r in range(3):
        qt[:, r] = R[r, 0] * p[:, 0] + R[r, 1] * p[:, 1] + R[r, 2] * p[:, 2] + t[r]
    n_in = int(np.floor(rho * n))
    order = np.argsort(_bits(seed, 4, m), kind="stable")
    inl = np.zeros(n, dtype=bool)
    inl[order[:n_in]] = True
    noise = gauss(seed, 5, idx3) * (tau / 3.0)
    lo, hi = qt.min(axis=0), qt.max(axis=0)
    q_out = lo[None, :] + uniform01(seed, 6, idx3) * (hi - lo)[None, :]
    q = np.where(inl[:, None], qt + noise, q_out)
    return Scene(p.astype(np.float32), q.astype(np.float32), R, t, inl)


def make_config_scene(name: str) -> tuple[Config, Scene]:
    cfg = CONFIGS[name]
    return cfg, make_scene(cfg.n, cfg.rho, cfg.L, cfg.tau, cfg.seed)


def rotation_error_deg(R: np.ndarray, R_gt: np.ndarray) -> float:
    c = (np.trace(np.asarray(R, dtype=np.float64) @ R_gt.T) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def stream_rho(cfg: Config, k: int) -> float:
    """Inlier ratio of frame k of a STREAM of distinct scenes of `cfg`'s shape: frame 0 is the config's own scene; frame k > 0 draws
    its ratio uniformly in [2/3, 4/3] of the config's (C2: 0.10 .. 0.20) from the counter hash of its seed — edge counts then move
    by ~1.6 x and triangle counts by ~4 x from frame to frame (a stream whose frames differ)."""
    if k == 0:
        return cfg.rho
    u = float(uniform01(cfg.seed + k, 9, np.arange(1, dtype=np.uint64))[0])
    return cfg.rho * (2.0 / 3.0 + (2.0 / 3.0) * u)


def make_stream_scenes(name: str, count: int) -> tuple[Config, list[Scene]]:
    """`count` distinct scenes of config `name`'s shape (same n, L, tau, parameters): seeds cfg.seed + k, inlier ratios stream_rho."""
    cfg = CONFIGS[name]
    return cfg, [make_scene(cfg.n, stream_rho(cfg, k), cfg.L, cfg.tau, cfg.seed + k) for k in range(count)]
