"""Independent float64 numpy restatement of the SAC-COT hot path (SURVEY.md §4 tier 1).

TEST INFRASTRUCTURE ONLY, and deliberately a *different program* from saccot_oracle.c: dense broadcasting for
the compatibility graph, boolean matrix algebra for the triangles, LAPACK SVD (numpy.linalg.svd) with the
textbook det fix for Kabsch, broadcasting for the inlier counts.  It shares no code and no arithmetic order
with the C restatement or the HIP kernels, so it can only agree with them to tolerance: decisions
(edge / inlier) are compared outside a guard band around the thresholds, values within a stated tolerance.

PARITY UNPINNED: there is no reference implementation to follow (/root/reference/README.md:1-2 is the whole
tree); this follows README.md:2 + BASELINE.json north_star + SURVEY.md §8(a).
"""
from __future__ import annotations

import numpy as np


def derive(sigma: float, t_cmp: float, tau: float):
    return sigma * np.sqrt(-2.0 * np.log(t_cmp)), -1.0 / (2.0 * sigma * sigma), tau * tau


def compat(src: np.ndarray, tgt: np.ndarray, sigma: float, t_cmp: float, min_len: float):
    """Returns S (n,n) f64, A (n,n) bool, margin (n,n) f64 = how far each pair is from flipping its decision
    (min over the three thresholds, in the units of the distance), deg (n,)."""
    p = np.asarray(src, dtype=np.float64)
    q = np.asarray(tgt, dtype=np.float64)
    dp = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))
    dq = np.sqrt(((q[:, None, :] - q[None, :, :]) ** 2).sum(-1))
    d = np.abs(dp - dq)
    d_thr, nis, _ = derive(sigma, t_cmp, 1.0)
    A = (d <= d_thr) & (dp >= min_len) & (dq >= min_len)
    np.fill_diagonal(A, False)
    S = np.where(A, np.exp(d * d * nis), 0.0)
    margin = np.minimum(np.abs(d - d_thr), np.minimum(np.abs(dp - min_len), np.abs(dq - min_len)))
    np.fill_diagonal(margin, np.inf)
    return S, A, margin, A.sum(1)


def triangles_all(S: np.ndarray, A: np.ndarray):
    """All 3-cliques i<j<k and their weights (s_ij + s_ik) + s_jk, via masked broadcasting row by row."""
    n = A.shape[0]
    out_i, out_j, out_k, out_w = [], [], [], []
    U = np.triu(A, 1)
    for i in range(n):
        js = np.nonzero(U[i])[0]
        if js.size < 2:
            continue
        sub = U[np.ix_(js, js)]                     # edges among the later neighbours of i
        a, b = np.nonzero(sub)
        if a.size == 0:
            continue
        j, k = js[a], js[b]
        out_i.append(np.full(j.shape, i)); out_j.append(j); out_k.append(k)
        out_w.append((S[i, j] + S[i, k]) + S[j, k])
    if not out_i:
        z = np.zeros(0, dtype=np.int64)
        return z, z, z, np.zeros(0)
    return np.concatenate(out_i), np.concatenate(out_j), np.concatenate(out_k), np.concatenate(out_w)


def triangle_count(A: np.ndarray) -> int:
    """trace(A^3)/6 by integer matrix algebra — a third, independent way to count 3-cliques."""
    Ai = A.astype(np.int64)
    return int(((Ai @ Ai) * Ai).sum() // 6)


def kabsch(P: np.ndarray, Q: np.ndarray):
    """Least-squares rigid transform Q ~ R P + t for (m,3) point sets (textbook SVD + det fix)."""
    P = np.asarray(P, dtype=np.float64); Q = np.asarray(Q, dtype=np.float64)
    pc, qc = P.mean(0), Q.mean(0)
    H = (P - pc).T @ (Q - qc)
    U, s, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    D = np.diag([1.0, 1.0, d if d != 0 else 1.0])
    R = Vt.T @ D @ U.T
    return R, qc - R @ pc, s


def residual2(src: np.ndarray, tgt: np.ndarray, R: np.ndarray, t: np.ndarray) -> np.ndarray:
    p = np.asarray(src, dtype=np.float64); q = np.asarray(tgt, dtype=np.float64)
    e = p @ np.asarray(R, dtype=np.float64).T + np.asarray(t, dtype=np.float64) - q
    return (e * e).sum(-1)
