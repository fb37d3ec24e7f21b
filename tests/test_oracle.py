"""CPU suite, part 1: the CPU restatement (oracle/saccot_oracle.c) against
  (a) the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py),
  (b) an independent float64 numpy restatement (oracle/saccot_fp64.py) — tolerance + guard bands,
  (c) synthetic ground truth and algebraic properties.
PARITY UNPINNED by the reference: /root/reference/README.md:1-2 holds no code, tests or vectors, so (a)-(c)
are this repo's own pins; the header of saccot_oracle.c and DESIGN.md say the same."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _kw(g):
    s, tc, tau, ml = (float(x) for x in g["params"])
    return dict(sigma=s, t_cmp=tc, tau=tau, min_len=ml, max_triangles=int(g["T"]), rank_mode=0)


@pytest.mark.parametrize("name", ["micro64", "c0"])
def test_oracle_reproduces_golden(O, name):
    import hashlib
    g = _load(name)
    kw = _kw(g)
    S, bits, deg = O.compat(g["src"], g["tgt"], kw["sigma"], kw["t_cmp"], kw["min_len"], kw["tau"])
    assert np.array_equal(bits, g["bits"]) and np.array_equal(deg, g["deg"])
    assert hashlib.sha256(S.tobytes()).digest() == g["S_sha256"].tobytes()
    tri, key, total = O.triangles(S, bits, deg, kw["max_triangles"], 0)
    assert total == int(g["tri_total"]) and np.array_equal(tri, g["tri"]) and np.array_equal(key, g["key"])
    Rt = O.kabsch3(g["src"], g["tgt"], tri)
    assert Rt.tobytes() == g["Rt"].tobytes()
    cnt = O.score(g["src"], g["tgt"], Rt, kw["tau"])
    assert np.array_equal(cnt, g["cnt"])
    res = O.register(g["src"], g["tgt"], threads=2, **kw)
    assert res["rc"] == 0 and res["best_rank"] == int(g["best_rank"]) and res["best_count"] == int(g["best_count"])
    assert res["R"].tobytes() == g["R"].tobytes() and res["t"].tobytes() == g["t"].tobytes()
    assert np.array_equal(res["mask"], g["mask"])
    assert res["edges"] == int(g["edges"])


def test_expf_against_numpy(O):
    xs = np.concatenate([-np.logspace(-8, np.log10(87.0), 4000), [0.0, -1e-30, -0.10536052]]).astype(np.float32)
    got = np.array([O.expf(x) for x in xs], dtype=np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.all(np.abs(got - ref) <= 2.5e-7 * ref)
    assert O.expf(0.0) == 1.0 and O.expf(-1000.0) == O.expf(-87.0) > 0.0


@pytest.mark.parametrize("n,seed,L,tau", [(300, 5, 1.0, 0.05), (500, 1000, 1.0, 0.05), (400, 9, 50.0, 0.6)])
def test_compat_against_fp64(pkg, O, n, seed, L, tau):
    from oracle import saccot_fp64 as F
    sc = pkg.synth.make_scene(n, 0.3, L, tau, seed)
    S, bits, deg = O.compat(sc.src, sc.tgt, tau, 0.9, tau, tau)
    S64, A64, margin, deg64 = F.compat(sc.src, sc.tgt, tau, 0.9, tau)
    A32 = S > 0
    # unpack the bit rows independently of S
    unpacked = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
    assert np.array_equal(unpacked, A32) and np.array_equal(deg, A32.sum(1))
    safe = margin > 1e-5 * L                     # guard band: pairs this close to a threshold may flip in fp32
    assert np.array_equal(A32[safe], A64[safe])
    assert (~safe).sum() - n < 1e-3 * n * n     # (diagonal margins are +inf)
    both = A32 & A64
    # weight tolerance vs exp() in float64: ~1 ulp of the polynomial + the fp32 rounding of two ~L-sized distances,
    # which enters s through d with slope <= 0.61/sigma
    assert np.abs(S[both] - S64[both]).max() < 2e-6 + 2e-7 * L / tau
    assert np.array_equal(S, S.T)


def test_triangles_against_numpy_enumeration(pkg, O):
    from oracle import saccot_fp64 as F
    sc = pkg.synth.make_scene(260, 0.4, 1.0, 0.05, 77)
    S, bits, deg = O.compat(sc.src, sc.tgt, 0.05, 0.9, 0.05, 0.05)
    A = S > 0
    i, j, k, _ = F.triangles_all(S.astype(np.float64), A)
    assert len(i) == F.triangle_count(A)
    for T in (1, 17, 1000, len(i), len(i) + 5):
        tri, key, total = O.triangles(S, bits, deg, T, 0)
        assert total == len(i) and len(tri) == min(T, len(i))
        w = ((S[i, j] + S[i, k]) + S[j, k]).astype(np.float32).view(np.uint32)   # fp32 adds, spec order
        order = np.lexsort((k, j, i, -w.astype(np.int64)))[:T]
        assert np.array_equal(key, w[order])
        assert np.array_equal(tri, np.stack([i[order], j[order], k[order]], 1))
    # degree ranking: integer keys, massive ties, same tie-break
    tri, key, _ = O.triangles(S, bits, deg, 500, 1)
    d = (deg[i].astype(np.int64) + deg[j] + deg[k])
    order = np.lexsort((k, j, i, -d))[:500]
    assert np.array_equal(key, d[order].astype(np.uint32))
    assert np.array_equal(tri, np.stack([i[order], j[order], k[order]], 1))


def test_kabsch_against_lapack_svd(pkg, O):
    from oracle import saccot_fp64 as F
    sc = pkg.synth.make_scene(400, 1.0, 1.0, 0.02, 3)
    rng = np.random.default_rng(0)
    tri = np.sort(np.stack([rng.choice(400, 3, replace=False) for _ in range(600)]), axis=1).astype(np.uint32)
    Rt = O.kabsch3(sc.src, sc.tgt, tri)
    worst_R = worst_t = 0.0
    checked = 0
    for h, (a, b, c) in enumerate(tri):
        R64, t64, s = F.kabsch(sc.src[[a, b, c]], sc.tgt[[a, b, c]])
        if s[1] < 0.05 * s[0]:
            continue                                   # near-collinear: ill-conditioned, excluded from the tolerance
        checked += 1
        worst_R = max(worst_R, np.abs(Rt[h, :9].reshape(3, 3) - R64).max())
        worst_t = max(worst_t, np.abs(Rt[h, 9:] - t64).max())
    assert checked > 400
    assert worst_R < 2e-4 and worst_t < 2e-4           # fp32 Jacobi vs fp64 LAPACK, unit-size scene
    R = Rt[:, :9].reshape(-1, 3, 3).astype(np.float64)
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5


def test_score_against_fp64_residuals(pkg, O):
    from oracle import saccot_fp64 as F
    cfg, sc = pkg.synth.make_config_scene("C0")
    g = _load("c0")
    Rt, tau = g["Rt"], cfg.tau
    cnt = O.score(sc.src, sc.tgt, Rt, tau)
    for h in range(len(Rt)):
        d2 = F.residual2(sc.src, sc.tgt, Rt[h, :9].reshape(3, 3), Rt[h, 9:])
        band = np.abs(d2 - tau * tau) <= 1e-4 * tau * tau
        lo = int((d2[~band] < tau * tau).sum())
        assert lo <= cnt[h] <= lo + int(band.sum())
    best = int(np.argmax(cnt))
    assert np.array_equal(O.mask(sc.src, sc.tgt, Rt[best], tau).sum(), cnt[best])


def test_best_key_tie_break(O):
    cnt = np.array([3, 9, 9, 0, 9], np.uint32)
    k = O.best_key(cnt)
    assert k >> 32 == 9 and 0xFFFFFFFF - (k & 0xFFFFFFFF) == 1        # ties -> best-ranked (lowest index)
    assert O.best_key(np.zeros(4, np.uint32)) == 0
    k = O.best_key(cnt, np.array([10, 7, 5, 1, 6], np.uint32))
    assert 0xFFFFFFFF - (k & 0xFFFFFFFF) == 5


@pytest.mark.parametrize("name,rot_tol", [("C0", 3.0), ("C1", 1.0)])
def test_known_answer_synthetic_ground_truth(pkg, O, name, rot_tol):
    cfg, sc = pkg.synth.make_config_scene(name)
    r = O.register(sc.src, sc.tgt, threads=4, **cfg.params())
    assert r["rc"] == 0
    assert pkg.synth.rotation_error_deg(r["R"], sc.R_gt) < rot_tol
    assert np.linalg.norm(r["t"] - sc.t_gt) < cfg.tau
    m = r["mask"].astype(bool)
    assert (m & sc.inlier).sum() >= 0.95 * sc.inlier.sum() and (m & ~sc.inlier).sum() <= 0.02 * cfg.n


def test_noise_free_all_inliers_recovers_ground_truth(pkg, O):
    sc = pkg.synth.make_scene(120, 1.0, 1.0, 1e-7, 5)
    r = O.register(sc.src, sc.tgt, 0.01, 0.9, 0.01, 0.01, 500)
    assert r["rc"] == 0 and r["mask"].all()
    # (arccos near 1 amplifies fp32 rounding to ~0.03 deg, so compare the matrices: the north-star 1e-5 bar)
    assert np.abs(r["R"] - sc.R_gt).max() < 1e-5 and np.abs(r["t"] - sc.t_gt).max() < 1e-5


def test_permutation_equivariance(pkg, O):
    cfg, sc = pkg.synth.make_config_scene("C0")
    perm = np.random.default_rng(1).permutation(cfg.n)
    a = O.register(sc.src, sc.tgt, threads=2, **cfg.params())
    b = O.register(sc.src[perm], sc.tgt[perm], threads=2, **cfg.params())
    assert a["tri_total"] == b["tri_total"] and a["edges"] == b["edges"]
    # the lexicographic tie-break depends on the labelling, so the two winners may differ — but only among
    # hypotheses of (nearly) the same quality
    assert abs(a["best_count"] - b["best_count"]) <= 3
    assert np.array_equal(O.mask(sc.src, sc.tgt, np.concatenate([b["R"].ravel(), b["t"]]), cfg.tau)[perm], b["mask"])


def test_errors_and_degenerate_inputs(pkg, O):
    cfg, sc = pkg.synth.make_config_scene("C0")
    assert O.register(sc.src[:2], sc.tgt[:2], **cfg.params())["rc"] == -1            # n < 3
    bad = sc.src.copy(); bad[3, 0] = np.inf
    assert O.register(bad, sc.tgt, **cfg.params())["rc"] == -1                         # non-finite
    assert O.register(sc.src, sc.tgt, **dict(cfg.params(), t_cmp=1.0))["rc"] == -1     # bad parameter
    r = O.register(sc.src[:40], (sc.src[:40] * 37).astype(np.float32), 0.001, 0.9, 0.001, 0.001, 50)
    assert r["rc"] == -5 and np.array_equal(r["R"], np.eye(3)) and not r["mask"].any()  # no triangle
    r = O.register(sc.src, sc.tgt, **dict(cfg.params(), max_triangles=10_000_000))
    assert r["rc"] == 0 and r["t_eff"] == r["tri_total"]                               # T > triangles -> T_eff


def test_refine_against_lapack(pkg, O):
    """so_refine (fp64 least-squares refit over the inlier mask) against numpy's SVD Kabsch."""
    from oracle import saccot_fp64 as F
    cfg, sc = pkg.synth.make_config_scene("C1")
    r = O.register(sc.src, sc.tgt, threads=4, **cfg.params())
    done, Rt = O.refine(sc.src, sc.tgt, r["mask"], np.concatenate([r["R"].ravel(), r["t"]]))
    m = r["mask"].astype(bool)
    R64, t64, _ = F.kabsch(sc.src[m], sc.tgt[m])
    assert done and np.abs(Rt[:9].reshape(3, 3) - R64).max() < 2e-7 and np.abs(Rt[9:] - t64).max() < 2e-7
    assert pkg.synth.rotation_error_deg(Rt[:9].reshape(3, 3), sc.R_gt) < pkg.synth.rotation_error_deg(r["R"], sc.R_gt)
    none, Rt_same = O.refine(sc.src, sc.tgt, np.zeros(cfg.n, np.uint8), np.arange(12, dtype=np.float32))
    assert not none and np.array_equal(Rt_same, np.arange(12, dtype=np.float32))     # < 3 inliers: untouched


@pytest.mark.parametrize("rank_mode", [0, 1])
def test_stage_b_is_thread_count_independent(pkg, O, rank_mode):
    """Stage B of the restatement runs its enumeration passes over the rows in parallel (OpenMP); the merged
    histograms / per-row prefixes must give the same ranked list for any thread count — incl. massive ties."""
    for sc, tau, T in ((pkg.synth.make_scene(600, 0.3, 1.0, 0.05, 3), 0.05, 5000),
                       (pkg.synth.make_scene(150, 1.0, 1.0, 1e-7, 21), 0.05, 3000)):       # all weights nearly equal
        S, bits, deg = O.compat(sc.src, sc.tgt, tau, 0.9, tau, tau)
        ref = O.triangles(S, bits, deg, T, rank_mode, threads=1)
        for th in (2, 3, 8):
            got = O.triangles(S, bits, deg, T, rank_mode, threads=th)
            assert got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_restatement_under_address_and_ub_sanitizers():
    """the CPU restatement runs clean under -fsanitize=address,undefined (oracle/Makefile `asan`), on the
    whole path for C0 with 1 and 4 threads and on the degenerate inputs, in a child process (the sanitizer runtime
    has to be the first library loaded)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    code = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as O
O._LIB = None
_build = O.build
O.build = lambda: os.path.join(os.path.dirname(_build()), "libsaccot_oracle_asan.so")
cfg, scene = pkg.synth.make_config_scene("C0")
a = O.register(scene.src, scene.tgt, threads=1, **cfg.params())
b = O.register(scene.src, scene.tgt, threads=4, **cfg.params())
assert a["rc"] == b["rc"] == 0 and a["best_rank"] == b["best_rank"] and np.array_equal(a["mask"], b["mask"])
kw = dict(cfg.params(), rank_mode=1, max_triangles=10**7)
c = O.register(scene.src, scene.tgt, threads=4, **kw)          # T > number of triangles, degree ranking
assert c["rc"] == 0 and c["t_eff"] == c["tri_total"]
rng = np.random.default_rng(5)
src = rng.uniform(-1, 1, (40, 3)).astype(np.float32)
d = O.register(src, (src * 37).astype(np.float32), threads=2, **cfg.params())   # no edge at all
assert d["rc"] == -5
done, Rt = O.refine(scene.src, scene.tgt, a["mask"], np.concatenate([a["R"].ravel(), a["t"]]))
assert done
print("asan-ok")
''' % root
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("mode", [1, 2])
def test_truncated_scores_against_fp64(pkg, O, mode):
    """SURVEY §8f-2 `score_mode`: the truncated-MSE / truncated-MAE scores are sums of floor(1024 * max(0, 1 - x)) with
    x = d2 / tau^2 resp. d / tau.  Against a float64 recomputation every term can differ by one unit at most (fp32
    rounding of the residual at a quantisation step), and only terms that are positive on either side."""
    sc = pkg.synth.make_scene(400, 0.4, 1.0, 0.05, 13)
    rng = np.random.default_rng(13)
    inl = np.nonzero(sc.inlier)[0]
    tri = np.sort(np.stack([rng.choice(inl if h % 2 else 400, 3, replace=False) for h in range(300)]), axis=1).astype(np.uint32)
    Rt = O.kabsch3(sc.src, sc.tgt, tri)
    tau = 0.05
    got = O.score(sc.src, sc.tgt, Rt, tau, score_mode=mode).astype(np.int64)
    cnt = O.score(sc.src, sc.tgt, Rt, tau).astype(np.int64)
    ok = np.isfinite(Rt).all(axis=1)
    R = Rt[:, :9].reshape(-1, 3, 3).astype(np.float64); t = Rt[:, 9:].astype(np.float64)
    e = np.einsum("hij,nj->hni", R, sc.src.astype(np.float64)) + t[:, None, :] - sc.tgt.astype(np.float64)[None]
    d2 = (e * e).sum(-1)
    x = d2 / tau ** 2 if mode == 1 else np.sqrt(d2) / tau
    ref = np.floor(1024 * np.clip(1 - x, 0, None)).sum(-1)
    pos = (x < 1 + 1e-4).sum(-1)
    assert np.all(np.abs(got[ok] - ref[ok]) <= pos[ok] + 1)
    assert np.all(got[ok] <= 1024 * cnt[ok]) and got[ok].max() > 20000 and not got[~ok].any()
    for kw in (dict(), dict(score_mode=mode)):                              # the whole path runs in the new mode too
        cfg, scene = pkg.synth.make_config_scene("C0")
        res = O.register(scene.src, scene.tgt, threads=2, **cfg.params(), **kw)
        assert res["rc"] == 0 and pkg.synth.rotation_error_deg(res["R"], scene.R_gt) < 3.0
