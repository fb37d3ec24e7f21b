"""CPU: the exact-arithmetic half of the Gram filter's error bound, restated in numpy (sac-cot_amd/csrc/sc_score.hip, header of the
Gram filter; gram_coef_block / gram_tile_block; sc_gramref.hpp).

The filter evaluates  s^2 |R_h p + t_h - q|^2  as a 48-slot dot product of fp16 halves in the frame of a reference motion.  Its
shell eps_h has two kinds of terms: what the MATRIX PIPE loses when it adds the products (GX_ACC: a measured model, probed at run
time on the GPU: tests/test_gpu_gram_guard.py) and what the REPRESENTATION loses even with exact addition — the dropped lo x lo
products and split remainders (GX_Q), the norm feature's two pieces (GX_NORM), the hypothesis' own R^T R - I (3 g vb Pn) — plus the
algebra of the frame itself.  This file checks the second kind, which no hardware enters: for random scenes, frames and hypotheses
(near the frame, far from it, slightly non-orthogonal), the EXACT sum of the kept fp16 products differs from the exact squared
residual by less than the representation terms of gram_eps.  (The constants are restated here on purpose: a change on one side
only must fail.)
"""
import numpy as np

GX_RS = 256.0
GX_Q = 7.5e-7
GX_NORM = 2.5e-7


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _small_rot(rng, ang):
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def _split(x):
    """x (float64 array) -> (hi, lo) fp16 values as float64: hi = fp16(fp32(x)), lo = fp16(fp32(x - hi)) — split2 in sc_score.hip."""
    hi = np.float32(x).astype(np.float16).astype(np.float64)
    lo = np.float32(x - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def _case(rng, n, near_frac, scale):
    p = rng.uniform(-1, 1, (n, 3)) * scale
    Rt, tt = _rot(rng), rng.uniform(-1, 1, 3) * scale
    q = p @ Rt.T + tt + rng.normal(size=(n, 3)) * 0.01 * scale
    out = rng.random(n) > near_frac
    q[out] = rng.uniform(-2, 2, (int(out.sum()), 3)) * scale
    p = np.float32(p).astype(np.float64); q = np.float32(q).astype(np.float64)      # the inputs are fp32
    # the frame: a rotation near the truth (what the vote finds), orthogonal to fp64 rounding
    R0 = _small_rot(rng, 0.01) @ Rt
    u, _, vt = np.linalg.svd(R0); R0 = u @ vt
    t0 = tt + rng.normal(size=3) * 0.01 * scale
    c = 0.5 * (p.max(0) + p.min(0))
    hP = np.maximum(p.max(0) - c, c - p.min(0))
    cQ = 0.5 * (q.max(0) + q.min(0)); hQ = np.maximum(q.max(0) - cQ, cQ - q.min(0))
    Qno = np.linalg.norm(hQ) + np.linalg.norm(cQ - t0 - R0 @ c)
    hmax = max(hP.max(), Qno)
    s = 2.0 ** (5 - int(np.floor(np.log2(hmax))))                                     # s hmax in [32, 64)
    Pn = s * np.linalg.norm(hP) * (1 + 2e-6); Qn = s * Qno * (1 + 2e-6)
    P = s * (p - c); Q = s * ((q - t0) @ R0 - c); V = Q - P                            # (R0^T dq)_i = sum_k R0[k][i] dq[k]
    assert np.abs(P).max() < 64 and np.abs(Q).max() < 64 and np.linalg.norm(Q, axis=1).max() <= Qn
    return dict(p=p, q=q, R0=R0, t0=t0, c=c, s=s, Pn=Pn, Qn=Qn, P=P, Q=Q, V=V, Rt=Rt, tt=tt, scale=scale)


def _check(rng, sc, R, t, label):
    s, c, R0, t0, P, Q, V, Pn, Qn = (sc[k] for k in ("s", "c", "R0", "t0", "P", "Q", "V", "Pn", "Qn"))
    R = np.float32(R).astype(np.float64); t = np.float32(t).astype(np.float64)       # a hypothesis is fp32
    dM = R0.T @ R - np.eye(3)
    G = R.T @ R - np.eye(3)
    Tp = s * (R0.T @ (t - t0) + dM @ c)
    # the exact value, two ways: in the original frame and in the filter's — the frame's algebra
    D_orig = (s ** 2) * np.sum((sc["p"] @ R.T + t - sc["q"]) ** 2, axis=1)
    D_star = np.sum((P @ dM.T + Tp - V) ** 2, axis=1)
    assert np.allclose(D_orig, D_star, rtol=1e-9, atol=1e-9 * (Pn + Qn) ** 2), label
    # features and coefficients as the kernels make them
    F = np.empty((len(P), 16))
    for i in range(3):
        for j in range(3):
            F[:, 3 * i + j] = Q[:, i] * P[:, j]
    F[:, 9] = 0.0
    F[:, 10:13] = 256.0 * P; F[:, 13:16] = 256.0 * V
    N = 0.5 * np.sum(V * V, axis=1)
    a = np.empty(16)
    a[:9] = GX_RS * (-2.0 * dM + G).ravel()
    a[9] = 2.0 * GX_RS
    a[10:13] = 2.0 * GX_RS / 256.0 * (dM.T @ Tp)
    a[13:16] = -2.0 * GX_RS / 256.0 * Tp
    Fh, Fl = _split(F); ah, al = _split(a)
    n0 = np.float32(N).astype(np.float16).astype(np.float64); n1 = np.float32(N - n0).astype(np.float16).astype(np.float64)
    Fh[:, 9], Fl[:, 9] = n0, n1
    al[9] = 0.0
    assert np.all(np.isfinite(Fh)) and np.all(np.isfinite(ah)) and np.abs(ah).max() * 16 < 65504, label   # fp16 range, alpha <= 16
    # MFMA 1: hi x hi; MFMA 2: coefficient hi x feature lo; MFMA 3: coefficient lo x feature hi — summed EXACTLY here
    D_split = (Fh @ ah + Fl @ ah + Fh @ al) / GX_RS + Tp @ Tp
    # the representation terms of gram_eps for the correspondences with |V'| <= vb
    Fn = np.linalg.norm(dM); g = np.abs(G).max(); Tn = np.linalg.norm(Tp)
    for vb in (0.25 * (Pn + Qn), Pn + Qn):
        sel = np.linalg.norm(V, axis=1) <= vb
        if not sel.any():
            continue
        Qb = min(Qn, Pn + vb)
        Sl = 2.0 * (Fn + 4.5 * g) * Qb * Pn + 2.0 * Fn * Tn * Pn + 2.0 * Tn * vb
        bound = GX_Q * Sl + GX_NORM * vb * vb + 3.0 * g * vb * Pn + 1e-6
        err = np.abs(D_split - D_star)[sel].max()
        assert err <= bound, (label, vb, err, bound, Fn, g, Tn)
    return np.abs(D_split - D_star).max()


def test_representation_terms_of_the_gram_bound_hold_in_exact_arithmetic():
    rng = np.random.default_rng(20260)
    worst = 0.0
    for it in range(40):
        sc = _case(rng, 400, float(rng.uniform(0.1, 0.9)), float(10.0 ** rng.uniform(-3, 3)))
        Rt, tt, scale = sc["Rt"], sc["tt"], sc["scale"]
        hyps = []
        for _ in range(6):      # near the frame
            hyps.append((_small_rot(rng, float(rng.uniform(0, 0.05))) @ Rt, tt + rng.normal(size=3) * 0.02 * scale))
        for _ in range(4):      # anywhere
            hyps.append((_rot(rng), rng.uniform(-1.5, 1.5, 3) * scale))
        for _ in range(3):      # slightly off a rotation (what a fp32 Kabsch returns, and worse: g up to 1e-4)
            hyps.append((_small_rot(rng, 0.02) @ Rt + rng.normal(size=(3, 3)) * float(10.0 ** rng.uniform(-7.5, -4.5)), tt))
        for k, (R, t) in enumerate(hyps):
            worst = max(worst, _check(rng, sc, R, t, (it, k)))
    assert worst > 0.0


def test_far_correspondences_of_a_near_hypothesis_are_outliers_by_the_triangle_inequality():
    """The cut: |V'| > reach_h + 1.05 st + 1e-3 with reach_h = |dM|_F Pn + |tau'|  =>  s |R_h p + t_h - q| > st."""
    rng = np.random.default_rng(20261)
    for it in range(40):
        sc = _case(rng, 600, 0.3, float(10.0 ** rng.uniform(-2, 2)))
        s, c, R0, t0, P, V, Pn = (sc[k] for k in ("s", "c", "R0", "t0", "P", "V", "Pn"))
        st = s * 0.02 * sc["scale"]
        R = np.float32(_small_rot(rng, float(rng.uniform(0, 0.03))) @ sc["Rt"]).astype(np.float64)
        t = np.float32(sc["tt"] + rng.normal(size=3) * 0.02 * sc["scale"]).astype(np.float64)
        dM = R0.T @ R - np.eye(3); Tp = s * (R0.T @ (t - t0) + dM @ c)
        reach = np.linalg.norm(dM) * Pn + np.linalg.norm(Tp)
        far = np.linalg.norm(V, axis=1) > reach + 1.05 * st + 1e-3
        resid = s * np.linalg.norm(sc["p"] @ R.T + t - sc["q"], axis=1)
        assert far.any() and np.all(resid[far] > 1.04 * st), (it, resid[far].min() / st)


def test_group_major_grid_order_is_a_bijection_and_spreads_the_working_workgroups_over_the_xcds():
    """score_gram_kernel's group-major decode of its one-dimensional grid (sc_score.hip; restated here on purpose):
    workgroup b -> row block groups - 1 - b / splits, split (b % splits + r) % splits with r = (b / splits * g / 8) % g,
    g = gcd(splits, 8).  Every (row block, split) is taken once, and for every ns the workgroups with split < ns — the
    ones of the near rows that work — fall on all eight XCDs (workgroup b runs on XCD b % 8) evenly."""
    from collections import Counter
    for splits in range(1, 9):
        g = min(splits & -splits, 8)
        for groups in (1, 7, 64, 196, 200):
            seen, per_xcd = set(), {ns: Counter() for ns in range(1, splits + 1)}
            for b in range(groups * splits):
                gi, si = divmod(b, splits)
                bx, by = groups - 1 - gi, (si + ((gi * g) >> 3) % g) % splits
                assert 0 <= bx < groups and 0 <= by < splits
                seen.add((bx, by))
                for ns in range(by + 1, splits + 1):
                    per_xcd[ns][b % 8] += 1
            assert len(seen) == groups * splits
            if groups >= 64:
                for ns, cnt in per_xcd.items():
                    v = [cnt[x] for x in range(8)]
                    assert max(v) - min(v) <= max(2, max(v) // 10), (splits, groups, ns, v)
