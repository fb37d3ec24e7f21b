#!/usr/bin/env python3
"""Generates tests/golden/*.npz — run from the repo root:  python tests/golden/make_golden.py

The reference (/root/reference/README.md:1-2) holds no fixtures, so these vectors are made by THIS repo:
inputs from the seeded generator (sac-cot_amd/synth.py), expected outputs from the CPU restatement
(oracle/saccot_oracle.c) — but only after this script has re-derived the decisive parts with independent
numpy code and found them identical:
  * adjacency bits vs a float64 broadcast computation, outside a guard band around the thresholds;
  * the number of 3-cliques vs trace(A^3)/6 in integer matrix algebra;
  * the ranked top-T list (keys AND order) vs a numpy enumeration + float32 adds + lexsort;
  * every inlier decision vs float64 residuals, outside a guard band around tau^2.
A fixture is data: inputs and expected outputs only.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def numpy_ranked(S32, A, T):
    from oracle import saccot_fp64 as F
    i, j, k, _ = F.triangles_all(S32.astype(np.float64), A)
    w = (S32[i, j] + S32[i, k]) + S32[j, k]            # float32 adds, same order as the spec
    key = w.astype(np.float32).view(np.uint32)
    order = np.lexsort((k, j, i, -key.astype(np.int64)))
    order = order[:T]
    return np.stack([i[order], j[order], k[order]], 1).astype(np.uint32), key[order]


def make(name, n, rho, L, tau, T, seed, pkg, O):
    from oracle import saccot_fp64 as F
    sc = pkg.synth.make_scene(n, rho, L, tau, seed)
    kw = dict(sigma=tau, t_cmp=0.9, tau=tau, min_len=tau, max_triangles=T, rank_mode=0)
    S, bits, deg = O.compat(sc.src, sc.tgt, kw["sigma"], kw["t_cmp"], kw["min_len"], kw["tau"])
    A32 = S > 0
    # -- independent checks before anything is written
    S64, A64, margin, _ = F.compat(sc.src, sc.tgt, kw["sigma"], kw["t_cmp"], kw["min_len"])
    safe = margin > 1e-5 * L
    assert np.array_equal(A32[safe], A64[safe]), "adjacency differs from fp64 outside the guard band"
    assert np.abs(S[A32 & A64] - S64[A32 & A64]).max() < 2e-6
    tri, key, total = O.triangles(S, bits, deg, T, 0)
    assert total == F.triangle_count(A32), "3-clique count differs from trace(A^3)/6"
    tri_np, key_np = numpy_ranked(S, A32, T)
    assert np.array_equal(tri, tri_np) and np.array_equal(key, key_np), "ranked list differs from numpy lexsort"
    Rt = O.kabsch3(sc.src, sc.tgt, tri)
    cnt = O.score(sc.src, sc.tgt, Rt, tau)
    for h in range(0, len(tri), max(1, len(tri) // 40)):
        d2 = F.residual2(sc.src, sc.tgt, Rt[h, :9].reshape(3, 3), Rt[h, 9:])
        ok = np.abs(d2 - tau * tau) > 1e-4 * tau * tau
        assert (d2[ok] < tau * tau).sum() <= cnt[h] <= (d2[ok] < tau * tau).sum() + (~ok).sum()
    res = O.register(sc.src, sc.tgt, threads=1, **kw)
    assert res["rc"] == 0 and res["best_count"] == cnt.max()
    out = dict(src=sc.src, tgt=sc.tgt, R_gt=sc.R_gt, t_gt=sc.t_gt, inlier=sc.inlier,
               params=np.array([kw["sigma"], kw["t_cmp"], kw["tau"], kw["min_len"]], np.float64),
               T=np.uint32(T), bits=bits, deg=deg,
               S_sha256=np.frombuffer(hashlib.sha256(S.tobytes()).digest(), np.uint8),
               tri=tri, key=key, tri_total=np.uint64(total), edges=np.uint64(int(deg.sum()) // 2), Rt=Rt, cnt=cnt,
               best_rank=np.uint32(res["best_rank"]), best_count=np.uint32(res["best_count"]),
               R=res["R"], t=res["t"], mask=res["mask"])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{path}: n={n} edges={int(deg.sum()) // 2} triangles={total} T_eff={len(tri)} "
          f"winner={res['best_rank']} inliers={res['best_count']} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    pkg = ge.load_package()
    O = ge.load_oracle()
    make("micro64", 64, 0.5, 1.0, 0.05, 50, 64, pkg, O)
    make("c0", 500, 0.30, 1.0, 0.05, 200, 1000, pkg, O)   # BASELINE.json configs[0]
